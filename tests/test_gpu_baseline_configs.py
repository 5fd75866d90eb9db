"""BASELINE.json configs[2], [3] and [4] at their STATED sizes on the GPU, and every A/B switch of libspt_hip.so.

The scenes of configs[3] / [4] (cfg4: GGX conductor + rough / smooth glass + 1024x512 EXR environment; cfg5: a 998 k-triangle
mesh in a homogeneous medium under an area light) are generated here from their seeds by scenes_amd/make_scenes.py
(numpy seeds 7 and 11, ~9 s; the 100 MB OBJ is not committed and does not travel to the GPU box).

What is compared with the CPU oracle is a strip of the image rendered with the full sample count (the oracle takes
seconds for 8 rows, hours for the image); the whole stated workload is rendered as well and tied to the strip through
properties that need no oracle: the strip's rows inside the big render are the same bits, a second run is the same bits.
"""
import os
import sys

import numpy as np
import pytest

import _util

pytestmark = pytest.mark.gpu

L1_TOL = 1e-3  # BASELINE.json north_star: per-pixel mean L1 < 1e-3


@pytest.fixture(scope="module")
def spt():
    return _util.load_pkg()


@pytest.fixture(scope="session")
def generated():
    """scenes_amd/generated/{cfg4_materials_env,cfg5_blob_medium}.json + assets (seeds 7 / 11 inside make_full)."""
    sys.path.insert(0, _util.SCENES)
    import make_scenes
    return make_scenes.make_full()


def _strip_vs_oracle(spt, sc, r, w, h, cam, rows, oracle_flags):
    """GPU and oracle render of the `rows` image rows in the middle of a w x h image (one shard of h / rows)."""
    count = h // rows
    index = count // 2
    got = r.render_shard(sc, spt.OutputConfig(w, h, None, cam), shard_index=index, shard_count=count, strip_rows=rows).copy()
    ref, _ = _util.oracle_render(sc, r, w, h, camera=cam, flags=oracle_flags, shard_index=index, shard_count=count, strip_rows=rows)
    return got, ref, spt.shard_rows(h, index, count, rows)


def test_cfg4_materials_env_1024x1024_at_512spp(spt, generated, monkeypatch):
    """configs[3]: microfacet conductor + dielectric glass + HDR .exr env-map MIS, 1024x1024 @ 512 spp, random sampler."""
    monkeypatch.delenv("SPT_REFERENCE_BVH", raising=False)
    sc = spt.load_scene(os.path.join(generated, "cfg4_materials_env.json"))
    r = spt.load_renderer(os.path.join(generated, "pt_random512.json"), seed=1)
    assert (r.spp, r.max_depth, r.sampler) == (512, 8, spt.SAMPLER_RANDOM)
    w = h = 1024
    got, ref, rows = _strip_vs_oracle(spt, sc, r, w, h, "main", 8, _util.ORACLE_EXHAUSTIVE)
    assert got.shape == (8, w, 3) and ref.max() > 1.0
    nan = np.isnan(ref)
    assert nan.mean() < 1e-3 and np.array_equal(nan, np.isnan(got))
    l1 = float(np.abs(got - ref)[~nan].mean())
    mism = int((got.view(np.uint32) != ref.view(np.uint32))[~nan].sum())
    assert l1 < L1_TOL and mism == 0, "strip of 8 rows @ 512 spp: %d words differ, L1 %.3g" % (mism, l1)
    # the whole stated workload: 537 M samples; the strip's rows inside it are the same bits, and so is a second run
    full = r.render_shard(sc, spt.OutputConfig(w, h, None, "main")).copy()
    st = r.last_stats
    assert st.samples == w * h * 512 and st.segments_closest + st.segments_shadow > 2 * st.samples   # ~2.5 ray segments per camera sample
    fn = np.isnan(full)
    assert fn.mean() < 1e-4      # sphere poles (sphere.rs:66): NaN in the reference too, see DESIGN.md
    assert np.array_equal(np.isnan(full[rows]), nan)
    assert np.array_equal(full[rows].view(np.uint32)[~nan], got.view(np.uint32)[~nan])
    again = r.render_shard(sc, spt.OutputConfig(w, h, None, "main"), samples_per_pass=96)
    assert np.array_equal(np.isnan(again), fn) and np.array_equal(again.view(np.uint32)[~fn], full.view(np.uint32)[~fn])
    assert 0.3 < float(full[~fn].mean()) < 30.0
    sc.close()


@pytest.mark.parametrize("bvh", ["own", "reference"])
def test_cfg5_million_triangles_medium_2048x2048_at_512spp_shard_0_of_8(spt, generated, bvh, monkeypatch):
    """configs[4]: 1 M-triangle mesh, deep SAH BVH + homogeneous medium, 2048x2048 @ 512 spp over 8 GPUs: one GPU's share.

    The oracle that can afford this scene walks the caller's trees (ORACLE_DEVICE; testing every ray against 10^6
    triangles is out of reach).  With SPT_REFERENCE_BVH=1 the kernels walk the same trees with the same slab arithmetic:
    bit-identical by construction.  By default the kernels walk their own padded trees, which never lose a hit the
    triangle test accepts, while an exact-box walk loses about one grazing ray in 1e7: the default
    mode is therefore held to the north-star L1 and to a handful of differing pixels out of 16 384."""
    if bvh == "reference":
        monkeypatch.setenv("SPT_REFERENCE_BVH", "1")
    else:
        monkeypatch.delenv("SPT_REFERENCE_BVH", raising=False)
    sc = spt.load_scene(os.path.join(generated, "cfg5_blob_medium.json"))
    assert sc.desc.n_tris > 990_000
    r = spt.load_renderer(os.path.join(generated, "pt_recurrence512.json"), seed=1)
    assert (r.spp, r.max_depth, r.sampler) == (512, 8, spt.SAMPLER_RECURRENCE)
    w = h = 2048
    got, ref, rows = _strip_vs_oracle(spt, sc, r, w, h, "main", 8, _util.ORACLE_DEVICE)
    assert np.isfinite(ref).all() and np.isfinite(got).all() and ref.max() > 0.2
    l1 = float(np.abs(got - ref).mean())
    bad_pixels = int((got.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
    assert l1 < L1_TOL, l1
    if bvh == "reference":
        assert bad_pixels == 0, "caller's trees, oracle's arithmetic: %d pixels differ" % bad_pixels
    else:
        assert bad_pixels <= 8, "%d of %d pixels differ (expected: the few rays an exact-box walk loses)" % (bad_pixels, 8 * w)
    if bvh == "own":
        # the stated per-GPU workload: shard 0 of 8 in 16-row strips = 256 rows, 268 M samples, ~3.8 ray segments each
        share = r.render_shard(sc, spt.OutputConfig(w, h, None, "main"), shard_index=0, shard_count=8, strip_rows=16).copy()
        st = r.last_stats
        assert share.shape == (256, w, 3) and st.samples == 256 * w * 512
        assert np.isfinite(share).all() and (st.segments_closest + st.segments_shadow) > 3 * st.samples
        # rows 1024 .. 1031 of the image belong to shard 0 of 8 (strip 64): the strip rendered above, bit for bit
        own_rows = spt.shard_rows(h, 0, 8, 16)
        pos = np.searchsorted(own_rows, rows)
        assert np.array_equal(own_rows[pos], rows)
        assert np.array_equal(share[pos].view(np.uint32), got.view(np.uint32))
        again = r.render_shard(sc, spt.OutputConfig(w, h, None, "main"), shard_index=0, shard_count=8, strip_rows=16, samples_per_pass=40)
        assert np.array_equal(again.view(np.uint32), share.view(np.uint32))
    sc.close()


def test_cfg3_cube_4096x4096_at_1024spp_shard_3_of_8(spt):
    """configs[2]: test_scene_01 at 4096x4096 @ 1024 spp tile-sharded over 8 GPUs: one GPU's share (2.1 G samples) through
    the closed forms of SURVEY 8c, and the sharding through equality with the same rows of a one-shard render."""
    sc = spt.load_scene(os.path.join(_util.SCENES, "cfg2_cube.json"))
    r = spt.load_renderer(os.path.join(_util.SCENES, "pt.json"), seed=1)
    w = h = 4096
    rows = spt.shard_rows(h, 3, 8, 16)
    r.spp = 1024
    share = r.render_shard(sc, spt.OutputConfig(w, h), shard_index=3, shard_count=8, strip_rows=16).copy()
    st = r.last_stats
    assert share.shape == (512, w, 3) and st.samples == 512 * w * 1024
    l = np.array([1.0, 1.0, 1.0]) / np.sqrt(3.0)
    c, s = np.cos(np.radians(60.0)), np.sin(np.radians(60.0))
    lum = [5.0 / np.pi * max(float(np.dot(n, l)), 0.0) for n in ([s, 0.0, c], [-c, 0.0, s])]   # 0.3363, 1.2552
    g = share[..., 0]
    assert np.array_equal(share[..., 0], share[..., 1]) and np.array_equal(share[..., 1], share[..., 2])
    near = lambda v: np.abs(g - v) < 3e-5
    interior = near(0.0) | near(lum[0]) | near(lum[1])
    assert interior.mean() > 0.997                       # silhouette pixels are 1 / 4 as frequent as at 1024^2
    assert abs(near(lum[0]).mean() + near(lum[1]).mean() - 0.1846) < 0.004      # interleaved strips: every shard sees the cube
    assert abs(float(g.mean()) - 0.11295) < 5e-4
    assert set(np.unique(spt.film_to_rgb8(share)[interior])) == {0, 85, 255}
    assert abs(st.primary_hits / st.samples - 0.1846) < 0.002
    again = r.render_shard(sc, spt.OutputConfig(w, h), shard_index=3, shard_count=8, strip_rows=16, samples_per_pass=100)
    assert np.array_equal(again.view(np.uint32), share.view(np.uint32))
    # sharding: one shard of 8 == the same rows of the whole image (reduced spp: the whole image is 8 x the work)
    r.spp = 16
    full = r.render_shard(sc, spt.OutputConfig(w, h)).copy()
    part = r.render_shard(sc, spt.OutputConfig(w, h), shard_index=3, shard_count=8, strip_rows=16)
    assert np.array_equal(part.view(np.uint32), full[rows].view(np.uint32))
    assert abs(float(full[..., 0].mean()) - 0.11295) < 5e-4
    sc.close()


# ---- every A/B switch of libspt_hip.so (DESIGN.md "Debug / A-B switches"): each changes which kernels run, none may
# change a film.  SPT_REFERENCE_BVH=1 is the mode that walks bvh.rs:262-283's trees as the caller built them.
SWITCHES = [
    {"SPT_REFERENCE_BVH": "1"},
    {"SPT_REFERENCE_BVH": "1", "SPT_NO_LDS_GEO": "1"},
    {"SPT_NO_FUSED": "1"},
    {"SPT_NO_LDS_TABLES": "1"},
    {"SPT_NO_LDS_GEO": "1"},
    {"SPT_NO_LDS_GEO": "1", "SPT_NO_DYN_SHADOW": "1"},
    {"SPT_NO_LDS_GEO": "1", "SPT_NO_DYN_EXTEND": "1"},
    {"SPT_NO_LDS_GEO": "1", "SPT_DYN_REFILL": "64", "SPT_DYN_STEPS": "1"},
    {"SPT_NO_LDS_GEO": "1", "SPT_NO_STREAM": "1"},                              # the refilling state-machine walkers of trace.h for every ray class
    {"SPT_NO_LDS_GEO": "1", "SPT_STREAM_MASK": "7"},                            # the streaming walker (stream.h) for primary, shadow and extension rays
    {"SPT_NO_LDS_GEO": "1", "SPT_STREAM_MASK": "7", "SPT_STREAM_IFIF": "0", "SPT_STREAM_ROUNDS": "2", "SPT_STREAM_REFILL": "64"},   # while-while
    {"SPT_NO_TAIL_LOOP": "1"},                                                  # fused scenes: one launch per bounce even when few paths are left
    {"SPT_NO_PIXEL_CULL": "1"},
    {"SPT_NO_OVERLAP": "1"},
    {"SPT_PRIMARY_CHUNKS": "1"},
    {"SPT_PRIMARY_CHUNKS": "5"},
    {"SPT_BVH_MAX_LEAF": "2"},
    {"SPT_BOX_BAND_BYTES": "200000"},
    {"SPT_NO_CLASS_QUEUES": "1"},                                               # one hit queue per shard instead of one per BxDF class
    {"SPT_NO_FUSED": "1", "SPT_NO_CLASS_QUEUES": "1"},
    {"SPT_NO_PACK_FIRST": "1"},                                                 # bounce-0 records as (slot) instead of (instance | sample, pixel)
    {"SPT_NO_ROW_SPANS": "1"},                                                  # no per-row screen-space spans in k_primary
    {"SPT_RESOLVE_BATCH": "32"},
    {"SPT_FLAT_BUDGET": "0"},                                                   # tree walkers for every scene (flat.h off)
    {"SPT_FLAT_BUDGET": "100000"},                                              # exhaustive loops for every LDS-resident scene
    {"SPT_FLAT_BUDGET": "100000", "SPT_NO_FUSED": "1"},
    {"SPT_NO_EYE_BLOB": "1"},                                                   # primary rays through the plain geometry (eye.h off)
]
ALL_SWITCHES = sorted({k for s in SWITCHES for k in s})


@pytest.mark.parametrize("switch", SWITCHES, ids=lambda s: "+".join("%s=%s" % kv for kv in sorted(s.items())))
@pytest.mark.parametrize("scene_name,camera", [("cfg2_cube.json", None), ("t_materials.json", "main"), ("t_medium.json", None), ("t_bezier.json", "low")])
def test_switch_sweep(spt, scene_name, camera, switch, monkeypatch):
    for k in ALL_SWITCHES:
        monkeypatch.delenv(k, raising=False)
    for k, v in switch.items():
        monkeypatch.setenv(k, v)
    sc = spt.load_scene(os.path.join(_util.SCENES, scene_name))
    flags = _util.device_oracle_flags()
    rays = _util.random_rays(sc, 50_000, seed=23)
    ref = _util.oracle_trace_closest(sc, rays, flags)
    got = sc.device_scene(0).trace_closest(rays)
    assert np.array_equal(ref["instance"] >= 0, got["instance"] >= 0)
    same_t = ref["t"].view(np.uint32) == got["t"].view(np.uint32)
    for f in ("instance", "prim"):
        assert np.array_equal(ref[f][same_t], got[f][same_t]), f
    # caller's trees + coincident surfaces (objects resting on the floor, rays from below): see test_gpu_parity
    assert same_t.all() if "SPT_REFERENCE_BVH" not in switch or not scene_name.startswith("t_") else same_t.mean() > 0.997
    wide = "SPT_BOX_BAND_BYTES" in switch
    r = spt.PathTracer(max_depth=8, sampler=spt.SAMPLER_RANDOM, spp=12, seed=5, filter_radius=1.5 if wide else 0.5)
    w, h = 128, 96
    want, _ = _util.oracle_render(sc, r, w, h, camera=camera, flags=flags)
    film = r.render_shard(sc, spt.OutputConfig(w, h, None, camera), samples_per_pass=5)
    assert np.isfinite(want).all() and want.max() > 0.05
    l1 = float(np.abs(film - want).mean())
    mism = int((film.view(np.uint32) != want.view(np.uint32)).sum())
    assert l1 < L1_TOL and mism == 0, "%s: %d words differ, L1 %.3g" % (switch, mism, l1)
    sc.close()
