#!/usr/bin/env python3
"""Debug driver for the streaming walker: small large-scene-path cases, one subprocess per configuration with a timeout."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, _util
spt = _util.load_pkg()
name, cam = sys.argv[1], (sys.argv[2] if sys.argv[2] != "-" else None)
sc = spt.load_scene(os.path.join(_util.SCENES, name))
r = spt.PathTracer(max_depth=4, sampler=spt.SAMPLER_RANDOM, spp=2, seed=3)
film = r.render_shard(sc, spt.OutputConfig(64, 48, None, cam))
want, _ = _util.oracle_render(sc, r, 64, 48, camera=cam, flags=_util.device_oracle_flags())
print(name, "mask", os.environ.get("SPT_STREAM_MASK"), "render words differ:", int((film.view(np.uint32) != want.view(np.uint32)).sum()), "of", film.size, flush=True)
''' % ROOT
for name, cam in (("cfg2_cube.json", "-"), ("t_materials.json", "main"), ("t_medium.json", "-")):
    for mask in ("1", "2", "4", "7"):
        env = dict(os.environ, SPT_NO_LDS_GEO="1", SPT_STREAM_MASK=mask)
        try:
            p = subprocess.run([sys.executable, "-c", CHILD, name, cam], env=env, timeout=40, capture_output=True, text=True)
            print(p.stdout.strip() or ("rc=%d " % p.returncode + p.stderr[-300:]), flush=True)
        except subprocess.TimeoutExpired:
            print(name, "mask", mask, "TIMEOUT", flush=True)
