#!/usr/bin/env python3
"""What ONE rank of an N-GPU weak-scaling run (bench.py: spp = 256 N, interleaved 16-row strips) costs on one GPU:
renders shard 0 of N for N = 1, 2, 4, 8 (always 268 M samples) and prints the step time.  Ideal weak scaling
needs these to be equal."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402

spt = _util.load_pkg()
scene = spt.load_scene(os.path.join(ROOT, "scenes_amd", "cfg2_cube.json"))
cfg = spt.OutputConfig(1024, 1024)
for n in (1, 2, 4, 8):
    r = spt.load_renderer(os.path.join(ROOT, "scenes_amd", "pt.json"), seed=1)
    r.spp = 256 * n
    for _ in range(2):
        r.render_shard(scene, cfg, shard_index=0, shard_count=n, strip_rows=16, reuse_output=True)
    t0 = time.perf_counter()
    k = 5
    for _ in range(k):
        r.render_shard(scene, cfg, shard_index=0, shard_count=n, strip_rows=16, reuse_output=True)
    ms = (time.perf_counter() - t0) / k * 1e3
    r.render_shard(scene, cfg, shard_index=0, shard_count=n, strip_rows=16, reuse_output=True, profile=True)
    st = r.last_stats
    kms = dict(zip(spt.KERNEL_NAMES, [round(x, 2) for x in st.kernel_ms]))
    print("N=%d chunks=%s ms/step %.3f  Gsamples/s per rank %.2f  %s" % (n, os.environ.get("SPT_PRIMARY_CHUNKS", "auto"), ms, st.samples / ms / 1e6, kms), flush=True)
