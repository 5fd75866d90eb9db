#!/usr/bin/env python3
"""Bezier-patch scenes (libspt_hip_bez.so) timed per kernel class: t_bezier.json (12 patches, LDS-resident) and
t_catmull.json (608 patch instances under a TLAS), 512^2 @ 64 spp.  GPU box: gpurun -- python3 tools/bezier_bench.py"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402

spt = _util.load_pkg()
for name, cam in (("t_bezier.json", "main"), ("t_catmull.json", "main"), ("t_materials.json", "main")):
    sc = spt.load_scene(os.path.join(ROOT, "scenes_amd", name))
    r = spt.PathTracer(max_depth=8, sampler=spt.SAMPLER_RANDOM, spp=64, seed=1)
    cfg = spt.OutputConfig(512, 512, None, cam)
    r.render_shard(sc, cfg, reuse_output=True)
    best = None
    for _ in range(3):
        r.render_shard(sc, cfg, reuse_output=True)
        st = r.last_stats
        if best is None or st.gpu_ms < best[0]:
            best = (st.gpu_ms, [st.kernel_ms[k] for k in range(spt.N_KERNELS)], st.samples, st.segments_closest + st.segments_shadow)
    ms, _, n, seg = best
    r.render_shard(sc, cfg, reuse_output=True, profile=True)      # per-class HIP-event times (one stream)
    per = [r.last_stats.kernel_ms[k] for k in range(spt.N_KERNELS)]
    classes = ", ".join("%s %.2f" % (spt.KERNEL_NAMES[k], per[k]) for k in range(spt.N_KERNELS) if per[k] > 0.005)
    print("%-16s 512^2 @ 64 spp: %8.2f ms = %7.1f Msamples/s, %.2f segments / sample; ms per class: %s" % (name, ms, n / ms / 1e3, seg / n, classes), flush=True)
