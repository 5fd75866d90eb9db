"""Film + BoxFilter of any radius (reference src/core/film.rs:47-92, src/filter/boxf.rs:5-34, pt.rs:278).

The reference keeps every sample with (offset - 0.5) and, per pixel, adds the UNWEIGHTED colours of all samples of
the (2R+1)^2 pixels around it (R = ceil(radius - 0.5)), rows then columns then sample order, and divides by the number
of those samples whose shifted offset lies within the radius.  The oracle's general film is checked here against an
independent numpy restatement (bit-exact at 1 spp, where the per-sample colours are the radius-0.5 film itself)."""
import ctypes as C
import os

import numpy as np
import pytest

import _util

spt = _util.load_pkg()
SCENE = os.path.join(_util.ROOT, "scenes_amd", "cfg2_cube.json")
W, H = 24, 18


def _offsets(seed, spp, sampler):
    lib = _util.oracle_lib()
    off = np.zeros((H, W, spp, 2), dtype=np.float32)
    for j in range(H):
        for i in range(W):
            px = j * W + i
            if sampler == spt.SAMPLER_RECURRENCE:
                buf = np.zeros(2 * spp, dtype=np.float32)
                lib.oracle_r2_offsets(px, spp, spp, buf.ctypes.data)
                off[j, i] = buf.reshape(spp, 2)
            else:
                for s in range(spp):
                    buf = np.zeros(2, dtype=np.float32)
                    lib.oracle_rng_stream(seed, px, s, 2, buf.ctypes.data)
                    off[j, i, s] = buf
    return off


def _weight_sums(off, radius):
    """weight_sum of filter_pixel for every pixel (exact: a count)."""
    R = int(np.ceil(np.float32(radius) - np.float32(0.5)))
    spp = off.shape[2]
    wsum = np.zeros((H, W), dtype=np.float32)
    rad = np.float32(radius)
    for y in range(H):
        for x in range(W):
            n = 0
            for dj in range(-R, R + 1):
                if not 0 <= y + dj < H:
                    continue
                for di in range(-R, R + 1):
                    if not 0 <= x + di < W:
                        continue
                    o = off[y + dj, x + di] - np.float32(0.5)
                    wx = np.float32(di) + o[:, 0]
                    wy = np.float32(dj) + o[:, 1]
                    n += int(np.count_nonzero((np.abs(wx) <= rad) & (np.abs(wy) <= rad)))
            wsum[y, x] = n
    return wsum, R


def _render(radius, spp, sampler, seed=3, **kw):
    sc = spt.load_scene(SCENE)
    r = spt.PathTracer(max_depth=4, sampler=sampler, spp=spp, seed=seed, filter_radius=radius)
    film, _ = _util.oracle_render(sc, r, W, H, **kw)
    return film


@pytest.mark.parametrize("radius", [0.3, 0.8, 1.5, 2.2, 0.0])
def test_one_sample_per_pixel_is_bit_exact_against_numpy(radius):
    base = _render(0.5, 1, spt.SAMPLER_RANDOM)                 # = the sample colours themselves
    off = _offsets(3, 1, spt.SAMPLER_RANDOM)
    wsum, R = _weight_sums(off, radius)
    want = np.zeros((H, W, 3), dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        for y in range(H):
            for x in range(W):
                c = np.zeros(3, dtype=np.float32)
                for dj in range(-R, R + 1):
                    if not 0 <= y + dj < H:
                        continue
                    for di in range(-R, R + 1):
                        if 0 <= x + di < W:
                            c = c + base[y + dj, x + di]
                want[y, x] = c * (np.float32(1.0) / wsum[y, x])
    got = _render(radius, 1, spt.SAMPLER_RANDOM)
    assert np.array_equal(np.isnan(want), np.isnan(got))
    ok = ~np.isnan(want)
    assert np.array_equal(want[ok].view(np.uint32), got[ok].view(np.uint32))
    if radius < 0.5:
        assert np.isnan(got).any() or np.isinf(got).any() or (wsum > 0).all()   # pixels with no sample inside divide by 0


@pytest.mark.parametrize("sampler,spp", [(spt.SAMPLER_RECURRENCE, 6), (spt.SAMPLER_RANDOM, 5)])
def test_many_samples_match_window_sums(sampler, spp):
    radius = 1.2
    base = _render(0.5, spp, sampler).astype(np.float64) * spp
    off = _offsets(3, spp, sampler)
    wsum, R = _weight_sums(off, radius)
    assert R == 1 and wsum.min() > 0 and len(np.unique(wsum)) > 3     # the weights really depend on the offsets
    want = np.zeros((H, W, 3))
    for y in range(H):
        for x in range(W):
            want[y, x] = base[max(0, y - R):y + R + 1, max(0, x - R):x + R + 1].sum(axis=(0, 1)) / wsum[y, x]
    got = _render(radius, spp, sampler)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-6)
    # unweighted colours over a (2R+1)^2 window but only the in-radius samples counted: brighter than the mean
    assert got.mean() > 1.5 * _render(0.5, spp, sampler).mean()


def test_jittered_sampler_and_shards_agree_with_the_full_film():
    sc = spt.load_scene(SCENE)
    r = spt.PathTracer(max_depth=3, sampler=spt.SAMPLER_JITTERED, spp=4, division_x=2, division_y=2, seed=9, filter_radius=1.5)
    full, st_full = _util.oracle_render(sc, r, W, H)
    for k in range(3):
        part, st = _util.oracle_render(sc, r, W, H, shard_index=k, shard_count=3, strip_rows=2)
        rows = spt.shard_rows(H, k, 3, 2)
        assert np.array_equal(part.view(np.uint32), full[rows].view(np.uint32))
        assert st.samples > len(rows) * W * 4          # halo rows are traced too
    assert st_full.samples == H * W * 4


def test_negative_radius_gives_the_reference_nan_film():
    got = _render(-0.7, 1, spt.SAMPLER_RANDOM)     # radius_int = -1: both loops empty, 0 * (1 / 0)
    assert np.isnan(got).all()


def test_renderer_file_radius_reaches_the_params(tmp_path):
    path = tmp_path / "pt.json"
    path.write_text('{"type": "pt", "max_depth": 3, "sampler": {"type": "random", "spp": 2}, "filter": {"type": "box", "radius": 1.5}}')
    r = spt.load_renderer(str(path))
    assert r.filter_radius == 1.5
    p = r.params(8, 8)
    assert p.flags & spt.RENDER_BOX_RADIUS and p.filter_radius == 1.5
    path.write_text('{"type": "pt", "max_depth": 3, "sampler": {"type": "random", "spp": 2}, "filter": {"type": "box", "radius": 0.5}}')
    p = spt.load_renderer(str(path)).params(8, 8)
    assert not (p.flags & spt.RENDER_BOX_RADIUS)
