#!/usr/bin/env python3
"""One rank's step time for its share of the metric's run (1024^2 @ 256 spp, cfg2) at world sizes 1, 2, 4, 8, measured on
ONE GPU: what strong scaling can reach before any second GPU is involved (fixed per-step costs: launches, syncs, D2H).
GPU box: gpurun -- python3 tools/shard_times.py"""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402

spt = _util.load_pkg()
sc = spt.load_scene(os.path.join(ROOT, "scenes_amd", "cfg2_cube.json"))
r = spt.load_renderer(os.path.join(ROOT, "scenes_amd", "pt.json"), seed=1)
r.spp = 256
cfg = spt.OutputConfig(1024, 1024)
t1 = None
worst_async = {}
for world in (1, 2, 4, 8):
    worst = 0.0
    for rank in sorted({0, world // 2, world - 1}):
        for _ in range(5):
            r.render_shard(sc, cfg, shard_index=rank, shard_count=world, strip_rows=16, reuse_output=True)
        t0 = time.perf_counter()
        n = 100
        for _ in range(n):
            r.render_shard(sc, cfg, shard_index=rank, shard_count=world, strip_rows=16, reuse_output=True)
        ms = (time.perf_counter() - t0) / n * 1e3
        worst = max(worst, ms)
        st = r.last_stats
        r.render_shard(sc, cfg, shard_index=rank, shard_count=world, strip_rows=16, reuse_output=True, profile=True)
        pst = r.last_stats
        classes = ", ".join("%s %d x %.3f" % (spt.KERNEL_NAMES[k], pst.kernel_launches[k], pst.kernel_ms[k] / max(pst.kernel_launches[k], 1))
                            for k in range(spt.N_KERNELS) if pst.kernel_launches[k])
        # the same steps queued without waiting for each film (SPT_RENDER_ASYNC), as bench.py times them
        t0 = time.perf_counter()
        for _ in range(n):
            r.render_shard(sc, cfg, shard_index=rank, shard_count=world, strip_rows=16, reuse_output=True, wait=False)
        r.wait(sc)
        ms_async = (time.perf_counter() - t0) / n * 1e3
        worst_async[world] = max(worst_async.get(world, 0.0), ms_async)
        print("world %d rank %d: %.3f ms / step wall (%.3f queued back to back), %.3f ms gpu (events); launches x avg ms: %s"
              % (world, rank, ms, ms_async, st.gpu_ms, classes), flush=True)
    t1 = t1 or worst
    print("world %d: slowest rank %.3f ms (%.3f queued) -> strong-scaling efficiency bound %.2f (%.2f queued)"
          % (world, worst, worst_async[world], t1 / (world * worst), worst_async[1] / (world * worst_async[world])), flush=True)
