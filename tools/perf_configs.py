#!/usr/bin/env python3
"""The `other_configs` block of bench.py on its own (BASELINE configs[3] = cfg4 and one GPU's share of configs[4] = cfg5):
Msamples/s, per-kernel-class ms, visit counters.  `python tools/perf_configs.py [cfg4|cfg5]`"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

only = sys.argv[1] if len(sys.argv) > 1 else ""
FEATURE_SCENES = {"bezier": ("t_bezier.json", "main"), "catmull": ("t_catmull.json", "main"), "pndf": ("t_pndf.json", "main"),
                  "subsurface": ("t_subsurface.json", None), "textured": ("t_textured.json", None), "materials": ("t_materials.json", "main")}
if only in FEATURE_SCENES:      # the feature scenes of scenes_amd/, 512^2 @ 64 spp (for tools/pmc_cfg.sh)
    spt = bench.load_pkg()
    name, cam = FEATURE_SCENES[only]
    sc = spt.load_scene(os.path.join(ROOT, "scenes_amd", name))
    r = spt.PathTracer(max_depth=8, sampler=spt.SAMPLER_RANDOM, spp=64, seed=1)
    for _ in range(2):
        r.render_shard(sc, spt.OutputConfig(512, 512, None, cam), reuse_output=True)
    ms = r.last_stats.gpu_ms
    r.render_shard(sc, spt.OutputConfig(512, 512, None, cam), reuse_output=True, profile=True)
    pst = r.last_stats
    print(only, round(ms, 2), "ms;", ", ".join("%s %.2f" % (spt.KERNEL_NAMES[k], pst.kernel_ms[k]) for k in range(spt.N_KERNELS) if pst.kernel_launches[k]))
    sys.exit(0)
res = bench.other_configs(bench.load_pkg(), 0, only=only)
for k, v in res.items():
    print(k, json.dumps(v))
