#!/usr/bin/env python3
"""BASELINE configs[3] and [4] at their stated sizes on one GPU: cfg4 1024^2 @ 512 spp (whole image) and the per-GPU
share of cfg5 (2048^2 @ 512 spp over 8 GPUs = shard 0 of 8)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402

spt = _util.load_pkg()
G = os.path.join(ROOT, "scenes_amd", "generated")
for name, scene, rend, cam, w, h, shard_count in (("cfg4", "cfg4_materials_env.json", "pt_random512.json", "main", 1024, 1024, 1),
                                                ("cfg5 (1/8)", "cfg5_blob_medium.json", "pt_recurrence512.json", "main", 2048, 2048, 8)):
    sc = spt.load_scene(os.path.join(G, scene))
    r = spt.load_renderer(os.path.join(G, rend), seed=1)
    cfg = spt.OutputConfig(w, h, None, cam)
    r.render_shard(sc, cfg, shard_index=0, shard_count=shard_count, reuse_output=True)
    t0 = time.perf_counter()
    r.render_shard(sc, cfg, shard_index=0, shard_count=shard_count, reuse_output=True)
    dt = time.perf_counter() - t0
    st = r.last_stats
    print("%s: %dx%d @ %d spp, %.1f M samples in %.1f ms = %.1f Msamples/s, %.2f segments/sample" %
          (name, w, h, r.spp, st.samples / 1e6, dt * 1e3, st.samples / dt / 1e6, (st.segments_closest + st.segments_shadow) / st.samples), flush=True)
