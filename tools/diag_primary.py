#!/usr/bin/env python3
"""Cost split of k_primary on cfg2: camera variants with 0 %, the stock 18 % and ~100 % primary hits."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402

spt = _util.load_pkg()
base = json.load(open(os.path.join(ROOT, "scenes_amd", "cfg2_cube.json")))
out_dir = os.path.join(ROOT, "scenes_amd", "generated")
os.makedirs(out_dir, exist_ok=True)
for name, eye, fwd in (("away", [0.0, 0.0, 7.0], [0.0, 0.0, 1.0]), ("stock", [0.0, 0.0, 7.0], [0.0, 0.0, -1.0]),
                       ("near", [0.0, 0.0, 2.6], [0.0, 0.0, -1.0]), ("off_axis", [3.0, 0.0, 7.0], [0.0, 0.0, -1.0])):
    sc = json.loads(json.dumps(base))
    sc["cameras"]["eye"], sc["cameras"]["forward"] = eye, fwd
    for k in ("primitives",):
        for p in sc[k]:
            if "obj_file" in p:
                p["obj_file"] = "../" + p["obj_file"]
    path = os.path.join(out_dir, "diag_%s.json" % name)
    json.dump(sc, open(path, "w"))
    scene = spt.load_scene(path)
    r = spt.load_renderer(os.path.join(ROOT, "scenes_amd", "pt.json"), seed=1)
    cfg = spt.OutputConfig(1024, 1024)
    r.render_shard(scene, cfg)
    r.render_shard(scene, cfg, profile=True)
    st = r.last_stats
    ms = dict(zip(spt.KERNEL_NAMES, [round(x, 3) for x in st.kernel_ms]))
    print(name, "hit_frac %.4f" % (st.primary_hits / st.samples), "gpu_ms %.3f" % st.gpu_ms, ms, flush=True)
