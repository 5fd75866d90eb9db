"""`debug_normal` (reference Cargo.toml:34-36, src/renderer/pt.rs:113-118): the colour of a path is `normal * 0.5 + 0.5`
of the first surface it reaches.  A second, radiance-independent pin of the oracle: the normal images of the two
shipped scenes have closed forms that involve no BxDF, no light and no RNG (sphere: (p - c) / r; cube: the three
columns of glam's from_rotation_y(60 deg)), so they check camera, NDC mapping, sampler offsets, instance transform and
the interpolated / re-projected normal on their own.  The GPU half compares device and oracle bit for bit."""
import os

import numpy as np
import pytest

import _util

spt = _util.load_pkg()


def _scene(name):
    return spt.load_scene(os.path.join(_util.SCENES, name))


def _pixel_dirs(w, h, spp, fov_deg=45.0):
    """float64 camera directions of every sample of every pixel of the R2 sampler (offsets from the oracle's own
    closed-form sampler seam, which tests/test_oracle_pins.py pins on its own), camera of the shipped scenes."""
    offs = np.zeros((h * w, spp, 2), dtype=np.float32)
    for p in range(h * w):
        buf = np.zeros(2 * spp, dtype=np.float32)
        _util.oracle_lib().oracle_r2_offsets(p, spp, spp, buf.ctypes.data)
        offs[p] = buf.reshape(spp, 2)
    offs = offs.reshape(h, w, spp, 2).astype(np.float64)
    j, i = np.mgrid[0:h, 0:w]
    x = ((i[..., None] + offs[..., 0]) / w - 0.5) * (w / h)
    y = ((h - j - 1)[..., None] + offs[..., 1]) / h - 0.5
    half_cot = 0.5 / np.tan(np.radians(fov_deg) / 2)
    d = np.stack([x, y, -half_cot * np.ones_like(x)], -1)
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def test_sphere_normal_image_closed_form():
    """scene 00: sphere r = 1 at (0.5, 0, 0), eye (0, 0, 5) looking down -z: colour = ((p - c) * 0.5 + 0.5) per sample."""
    w = h = 48
    spp = 4
    sc = _scene("cfg1_sphere.json")
    r = spt.PathTracer(max_depth=8, spp=spp, seed=1, debug_normal=True)
    film, _ = _util.oracle_render(sc, r, w, h)
    d = _pixel_dirs(w, h, spp)
    o = np.array([0.0, 0.0, 5.0]) - np.array([0.5, 0.0, 0.0])
    b = d @ o
    disc = b * b - (o @ o - 1.0)
    hit = disc > 0
    t = -b - np.sqrt(np.where(hit, disc, 0.0))
    n = o + d * t[..., None]
    col = np.where(hit[..., None], n * 0.5 + 0.5, 0.0)
    want = col.mean(axis=2)
    safe = (np.abs(disc) > 1e-3).all(axis=2)          # every sample clearly inside or clearly outside the silhouette
    assert safe.mean() > 0.9 and hit.all(axis=2).sum() > 150
    assert np.abs(film[safe] - want[safe]).max() < 3e-5    # f32 roots near the silhouette: sqrt of a small discriminant
    well = (np.abs(disc) > 0.05).all(axis=2)
    assert well.mean() > 0.8 and np.abs(film[well] - want[well]).max() < 2e-6
    assert np.abs(film - want).max() < 0.26            # a silhouette pixel can differ by one sample of four


def test_cube_normal_image_closed_form():
    """scene 01: unit cube [-1, 1]^3 rotated 60 degrees about Y; only the faces +z' and -x' face the eye at (0, 0, 7):
    normals (sin 60, 0, cos 60) and (-cos 60, 0, sin 60) (glam from_rotation_y columns; the radiance pin of
    test_oracle_pins.py rests on the same convention through n . l, this one reads the components themselves)."""
    w = h = 64
    sc = _scene("cfg2_cube.json")
    r = spt.PathTracer(max_depth=8, spp=2, seed=1, debug_normal=True)
    film, _ = _util.oracle_render(sc, r, w, h)
    c, s = np.cos(np.radians(60.0)), np.sin(np.radians(60.0))
    front, left = np.array([s, 0.0, c]) * 0.5 + 0.5, np.array([-c, 0.0, s]) * 0.5 + 0.5
    px = film.reshape(-1, 3).astype(np.float64)
    is_front = np.abs(px - front).max(axis=1) < 2e-6
    is_left = np.abs(px - left).max(axis=1) < 2e-6
    is_bg = (px == 0.0).all(axis=1)
    assert (is_front | is_left | is_bg).mean() > 0.95          # the rest are silhouette / edge pixels (mixtures)
    assert abs((is_front | is_left).mean() - 0.1846) < 0.02     # the hit fraction of SURVEY 8c
    assert is_left.sum() > is_front.sum() > 100
    # the same image with shading on is NOT the normal image (the flag is what changed it)
    lit, _ = _util.oracle_render(sc, spt.PathTracer(max_depth=8, spp=2, seed=1), w, h)
    assert np.abs(lit - film).max() > 0.2


def test_first_surface_after_a_medium_replaces_what_the_path_gathered():
    """pt.rs:116 ASSIGNS final_color: in-scattered light gathered inside a medium before the surface is dropped.  Every
    pixel of a debug film therefore lies in [0, 1] (or is the environment / black), whatever the lights are."""
    sc = _scene("t_medium.json")
    r = spt.PathTracer(max_depth=6, sampler=spt.SAMPLER_RANDOM, spp=4, seed=3, debug_normal=True)
    film, _ = _util.oracle_render(sc, r, 64, 48)
    lit, _ = _util.oracle_render(sc, spt.PathTracer(max_depth=6, sampler=spt.SAMPLER_RANDOM, spp=4, seed=3), 64, 48)
    assert np.isfinite(film).all() and film.min() >= 0.0 and film.max() <= 1.0 + 1e-6
    assert lit.max() > 1.5      # the lit image is not bounded like that


@pytest.mark.gpu
@pytest.mark.parametrize("scene_name,camera,sampler,size,spp", [
    ("cfg1_sphere.json", None, "recurrence", (96, 64), 16), ("cfg2_cube.json", None, "recurrence", (128, 128), 16),
    ("t_materials.json", "main", "random", (96, 64), 8), ("t_medium.json", None, "random", (96, 64), 8),
    ("t_textured.json", None, "random", (64, 48), 4), ("t_subsurface.json", None, "random", (64, 48), 4),
    ("t_bezier.json", "main", "random", (64, 48), 4)])
def test_gpu_debug_normal_matches_oracle(scene_name, camera, sampler, size, spp):
    sc = _scene(scene_name)
    r = spt.PathTracer(max_depth=6, sampler=spt.SAMPLER_RANDOM if sampler == "random" else spt.SAMPLER_RECURRENCE, spp=spp, seed=5,
                       debug_normal=True)
    cfg = spt.OutputConfig(size[0], size[1], used_camera_name=camera)
    want, _ = _util.oracle_render(sc, r, size[0], size[1], camera=camera, flags=_util.device_oracle_flags())
    for kw in ({}, {"samples_per_pass": 3}, {"shard_index": 1, "shard_count": 2, "strip_rows": 4}):
        got = r.render_shard(sc, cfg, **kw)
        ref = want
        if "shard_index" in kw:
            ref = want[spt.shard_rows(size[1], 1, 2, 4)]
        diff = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
        assert diff == 0, (scene_name, kw, diff)
    assert want.max() <= 1.0 + 1e-6 or sc.desc.env.width != 0
    lit = spt.PathTracer(max_depth=6, sampler=r.sampler, spp=spp, seed=5).render_shard(sc, cfg)
    assert not np.array_equal(lit, want)
