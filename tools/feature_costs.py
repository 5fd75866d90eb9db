#!/usr/bin/env python3
"""What the round's later features cost: cfg2 with the box filter at radius 0.5 / 0.3 / 1.5 / 2.5, and the Bezier-patch
scene (served by libspt_hip_bez.so) next to a same-size patch-free scene.  GPU box: gpurun -- python tools/feature_costs.py"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402

spt = _util.load_pkg()


def timed(sc, r, w, h, cam=None):
    cfg = spt.OutputConfig(w, h, None, cam)
    r.render_shard(sc, cfg, reuse_output=True)
    best = 1e30
    for _ in range(3):
        r.render_shard(sc, cfg, reuse_output=True)
        best = min(best, r.last_stats.gpu_ms)
    return best, r.last_stats.samples


sc = spt.load_scene(os.path.join(ROOT, "scenes_amd", "cfg2_cube.json"))
for radius in (0.5, 0.3, 1.5, 2.5):
    r = spt.PathTracer(max_depth=8, sampler=spt.SAMPLER_RECURRENCE, spp=64, seed=1, filter_radius=radius)
    ms, n = timed(sc, r, 1024, 1024)
    print("cfg2 1024^2 @ 64 spp, box radius %.1f: %7.2f ms, %6.1f M camera samples traced (halo included), %7.1f Msamples/s" % (radius, ms, n / 1e6, n / ms / 1e3), flush=True)
for name, cam in (("t_bezier.json", "main"), ("t_materials.json", "main"), ("t_textured.json", None)):
    s2 = spt.load_scene(os.path.join(ROOT, "scenes_amd", name))
    r = spt.PathTracer(max_depth=8, sampler=spt.SAMPLER_RANDOM, spp=64, seed=1)
    ms, n = timed(s2, r, 512, 512, cam)
    st = r.last_stats
    print("%-18s 512^2 @ 64 spp: %7.2f ms = %7.1f Msamples/s, %.2f segments / sample" % (name, ms, n / ms / 1e3, (st.segments_closest + st.segments_shadow) / n), flush=True)
