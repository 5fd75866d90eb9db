#!/usr/bin/env python3
"""Debug helper: locate non-finite pixels of a GPU render and the first sample index that produces them."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util
spt = _util.load_pkg()
scene = spt.load_scene(os.path.join(ROOT, sys.argv[1]))
cam = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
w = h = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
spp = int(sys.argv[4]) if len(sys.argv) > 4 else 128
r = spt.PathTracer(max_depth=8, sampler=spt.SAMPLER_RANDOM, spp=spp, seed=1)
film = r.render_shard(scene, spt.OutputConfig(w, h, None, cam))
bad = np.argwhere(~np.isfinite(film).all(-1))
print("non-finite pixels:", len(bad), bad[:8].tolist())
out = []
for j, i in bad[:4]:
    lo, hi = 0, spp
    while hi - lo > 1:   # first k with NaN among samples [0, k)
        mid = (lo + hi) // 2
        rr = spt.PathTracer(max_depth=8, sampler=spt.SAMPLER_RANDOM, spp=mid, seed=1)
        f = rr.render_shard(scene, spt.OutputConfig(w, h, None, cam), shard_index=int(j), shard_count=h, strip_rows=1)
        if np.isfinite(f[0, i]).all():
            lo = mid
        else:
            hi = mid
    out.append({"row": int(j), "col": int(i), "first_bad_sample": hi - 1})
print(json.dumps(out))
