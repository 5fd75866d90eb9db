#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (reference-faithful mode: divide slab test,
deterministic math).  The Rust reference cannot be run here, so these vectors pin the oracle against
regressions and give the GPU tests fixed inputs/outputs that travel to the GPU box.
Usage: python tests/make_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _util

CASES = [  # name, scene, camera, sampler, spp, (w, h), seed, box filter radius
    ("film_cfg1_sphere", "cfg1_sphere.json", None, "recurrence", 16, (64, 64), 1, 0.5),
    ("film_cfg2_cube", "cfg2_cube.json", None, "recurrence", 16, (64, 64), 1, 0.5),
    ("film_t_materials", "t_materials.json", "main", "random", 16, (64, 48), 1, 0.5),
    ("film_t_power_is", "t_power_is.json", "top", "jittered", 16, (64, 48), 1, 0.5),
    ("film_t_medium", "t_medium.json", None, "random", 16, (64, 48), 1, 0.5),
    ("film_t_plastic", "t_plastic.json", None, "random", 16, (64, 48), 1, 0.5),
    ("film_t_textured", "t_textured.json", None, "recurrence", 16, (64, 48), 1, 0.5),
    ("film_t_gltf", "t_gltf.gltf", "cam", "random", 16, (64, 48), 1, 0.5),
    ("film_t_subsurface", "t_subsurface.json", None, "random", 16, (64, 48), 1, 0.5),
    ("film_t_bezier", "t_bezier.json", "main", "random", 16, (64, 48), 1, 0.5),          # CubicBezier patches (bezier.rs)
    ("film_t_catmull", "t_catmull.json", "main", "random", 16, (64, 48), 1, 0.5),        # Catmull-Clark surfaces as Bezier patch instances (catmull.rs)
    ("film_t_pndf", "t_pndf.json", "main", "random", 16, (64, 48), 1, 0.5),              # P-NDF glints (pndf_conductor.rs, pndf_plastic.rs)
    ("film_cfg2_cube_box1p2", "cfg2_cube.json", None, "random", 8, (64, 64), 1, 1.2),   # film.rs:71-92 with radius_int = 1
]


def main():
    _util.ensure_cpu_build()
    spt = _util.load_pkg()
    kinds = {"random": spt.SAMPLER_RANDOM, "recurrence": spt.SAMPLER_RECURRENCE, "jittered": spt.SAMPLER_JITTERED}
    os.makedirs(_util.GOLDEN, exist_ok=True)
    for name, scene, cam, sampler, spp, (w, h), seed, radius in CASES:
        sc = spt.load_scene(os.path.join(_util.SCENES, scene))
        r = spt.PathTracer(max_depth=8, sampler=kinds[sampler], spp=spp, division_x=4, division_y=4, seed=seed, filter_radius=radius)
        film, st = _util.oracle_render(sc, r, w, h, camera=cam, flags=0)
        rays = _util.random_rays(sc, 4096, seed=17)
        hits = _util.oracle_trace_closest(sc, rays, 0)
        rays_any = rays.copy()
        rays_any["t_max"] = np.where(hits["instance"] >= 0, hits["t"] * np.float32(1.25), np.float32(6.0)).astype(np.float32)
        rays_any["t_max"][::2] *= np.float32(0.5)
        occ = _util.oracle_trace_any(sc, rays_any, 0)
        np.savez_compressed(os.path.join(_util.GOLDEN, name + ".npz"), film=film, rays=rays, hits=hits, rays_any=rays_any, occ=occ,
                            meta=np.array([w, h, spp, seed, kinds[sampler]], dtype=np.int64),
                            counters=np.array([st.segments_closest, st.segments_shadow, st.node_tests, st.tri_tests], dtype=np.int64))
        print(name, "mean", film.mean(), "hit rate", (hits["instance"] >= 0).mean())


if __name__ == "__main__":
    main()
