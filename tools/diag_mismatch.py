#!/usr/bin/env python3
"""Where (and by how much) the default device-built-BVH film differs from the oracle, and whether the
SPT_REFERENCE_BVH=1 film does."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402

spt = _util.load_pkg()
name, cam = (sys.argv[1], sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None) if len(sys.argv) > 1 else ("t_gltf.gltf", "cam")
w, h, spp, seed = 384, 288, 32, 77
r = spt.PathTracer(max_depth=8, sampler=spt.SAMPLER_RECURRENCE, spp=spp, seed=seed)
sc = spt.load_scene(os.path.join(ROOT, "scenes_amd", name))
ref, _ = _util.oracle_render(sc, r, w, h, camera=cam, flags=_util.ORACLE_DEVICE)
brute, _ = _util.oracle_render(sc, r, w, h, camera=cam, flags=_util.ORACLE_BRUTE_FORCE | _util.ORACLE_TIE_MIN_ID)
print("oracle tree vs oracle brute force: pixels differing:", int(((ref.view(np.uint32) != brute.view(np.uint32)).any(axis=2) & ~np.isnan(ref).any(axis=2)).sum()))
for mode in ("own", "reference"):
    if mode == "reference":
        os.environ["SPT_REFERENCE_BVH"] = "1"
    s2 = spt.load_scene(os.path.join(ROOT, "scenes_amd", name))
    got = r.render_shard(s2, spt.OutputConfig(w, h, None, cam), samples_per_pass=13)
    bad = np.argwhere((got.view(np.uint32) != ref.view(np.uint32)).any(axis=2) & ~np.isnan(ref).any(axis=2))
    bad_b = np.argwhere((got.view(np.uint32) != brute.view(np.uint32)).any(axis=2) & ~np.isnan(brute).any(axis=2))
    print(mode, "pixels differing from the tree oracle:", len(bad), " from the brute-force oracle:", len(bad_b))
    for y, x in bad[:5]:
        print("  pixel", (int(x), int(y)), "gpu", got[y, x], "oracle", ref[y, x], "delta*spp", (got[y, x] - ref[y, x]) * spp)
