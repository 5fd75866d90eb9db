"""Ad-hoc GPU diagnostics (not a test): prints where the HIP path and the oracle differ."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _util

spt = _util.load_pkg()
_util.ensure_cpu_build()
for name in sys.argv[1:] or ["cfg1_sphere.json", "cfg2_cube.json"]:
    sc = spt.load_scene(os.path.join(_util.SCENES, name))
    rays = _util.random_rays(sc, 200_000, seed=11)
    ref = _util.oracle_trace_closest(sc, rays, _util.ORACLE_SLAB_RECIPROCAL)
    got = sc.device_scene(0).trace_closest(rays)
    for f in ("instance", "prim", "t", "v", "w"):
        a, b = ref[f], got[f]
        bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
        print(name, f, "mismatch", len(bad), "of", len(a))
        if len(bad) and f in ("t", "v", "w"):
            ulps = np.abs(a.view(np.int32)[bad].astype(np.int64) - b.view(np.int32)[bad].astype(np.int64))
            print("   ulp diff: max", ulps.max(), "hist", np.bincount(np.minimum(ulps, 8)))
            k = bad[0]
            print("   e.g. ray", rays[k], "ref", ref[k], "got", got[k])
