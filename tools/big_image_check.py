#!/usr/bin/env python3
"""BASELINE configs[2] shape: test_scene_01 at 4096x4096, one shard of 8 (and the whole image at a lower spp):
closed-form face radiances, shard == the same rows of the full image."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402

spt = _util.load_pkg()
sc = spt.load_scene(os.path.join(ROOT, "scenes_amd", "cfg2_cube.json"))
r = spt.load_renderer(os.path.join(ROOT, "scenes_amd", "pt.json"), seed=1)
r.spp = 32
cfg = spt.OutputConfig(4096, 4096)
t0 = time.perf_counter()
full = r.render_shard(sc, cfg).copy()
t1 = time.perf_counter()
print("full 4096^2 @ %d spp: %.1f ms, %.1f Gsamples/s" % (r.spp, (t1 - t0) * 1e3, r.last_stats.samples / (t1 - t0) / 1e9))
g = full[..., 0]
l = np.array([1.0, 1.0, 1.0]) / np.sqrt(3.0)
c, s = np.cos(np.radians(60.0)), np.sin(np.radians(60.0))
lum = [5.0 / np.pi * max(float(np.dot(n, l)), 0.0) for n in ([s, 0.0, c], [-c, 0.0, s])]
near = lambda v: np.abs(g - v) < 1e-4
inside = near(0.0) | near(lum[0]) | near(lum[1])
print("interior fraction %.4f, lit fraction %.4f (expect 0.1846), mean %.5f (expect 0.11295)" % (inside.mean(), (near(lum[0]) | near(lum[1])).mean(), g.mean()))
assert inside.mean() > 0.995 and abs(float(g.mean()) - 0.11295) < 5e-4
rows = spt.shard_rows(4096, 3, 8, 16)
t0 = time.perf_counter()
shard = r.render_shard(sc, cfg, shard_index=3, shard_count=8, strip_rows=16)
t1 = time.perf_counter()
print("shard 3/8: %.1f ms" % ((t1 - t0) * 1e3))
assert np.array_equal(shard.view(np.uint32), full[rows].view(np.uint32))
r.spp = 1024
t0 = time.perf_counter()
shard = r.render_shard(sc, cfg, shard_index=3, shard_count=8, strip_rows=16)
t1 = time.perf_counter()
print("shard 3/8 @ 1024 spp (configs[2] per-GPU share, %.2f Gsamples): %.1f ms, %.1f Gsamples/s" % (r.last_stats.samples / 1e9, (t1 - t0) * 1e3, r.last_stats.samples / (t1 - t0) / 1e9))
gs = shard[..., 0]
ok = (np.abs(gs) < 1e-4) | (np.abs(gs - lum[0]) < 1e-4) | (np.abs(gs - lum[1]) < 1e-4)
assert ok.mean() > 0.995
print("ok")
