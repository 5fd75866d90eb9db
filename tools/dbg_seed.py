#!/usr/bin/env python3
"""One fuzz seed of tools/fuzz_scenes.py under explicit switch sets: which pixels differ from which oracle.
usage (GPU box): python3 tools/dbg_seed.py SEED [SIZE_MUL SPP_MUL]"""
import json
import os
import sys

os.environ.update(FUZZ_V3="1", FUZZ_V2="1")
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fuzz_scenes as fz
import _util

spt = _util.load_pkg()
seed = int(sys.argv[1])
size_mul, spp_mul = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, 1)
work = fz.stage_assets()
rng = np.random.default_rng(1000 + seed)
scene = fz.make_scene(rng, work)
path = os.path.join(work, "fuzz_%d.json" % seed)
json.dump(scene, open(path, "w"))
sampler = int(rng.integers(0, 3)); dx, dy = int(rng.integers(1, 4)), int(rng.integers(1, 4))
spp = dx * dy if sampler == spt.SAMPLER_JITTERED else int(rng.integers(1, 9))
radius = float(rng.choice([0.5, 0.5, 0.5, 0.3, 1.2, 1.6]))
r = spt.PathTracer(max_depth=int(rng.integers(1, 9)), sampler=sampler, spp=spp, division_x=dx, division_y=dy, seed=int(rng.integers(0, 1 << 30)), filter_radius=radius)
w, h = int(rng.integers(17, 120)) * size_mul, int(rng.integers(9, 90)) * size_mul
if sampler != spt.SAMPLER_JITTERED and spp_mul > 1:
    r.spp = spp * spp_mul
print("image %dx%d spp %d depth %d sampler %d radius %.1f" % (w, h, r.spp, r.max_depth, sampler, radius), flush=True)
refs = {}
sc = spt.load_scene(path)
print("patches", sc.desc.n_bezier_patches, "instances", sc.desc.n_instances)
for name, flags in (("exhaustive", _util.device_oracle_flags()), ("tree-walking", _util.ORACLE_DEVICE)):
    refs[name], _ = _util.oracle_render(sc, r, w, h, flags=flags)
sc.close()
SETS = ({}, {"SPT_BEZ_LDS": "1"}, {"SPT_STREAM_MASK": "0"}, {"SPT_STREAM_MASK": "2"}, {"SPT_STREAM_MASK": "4"}, {"SPT_STREAM_MASK": "1"}, {"SPT_STREAM_IFIF": "0"},
        {"SPT_REFERENCE_BVH": "1"})
for switches in SETS:
    for k in ("SPT_BEZ_LDS", "SPT_STREAM_MASK", "SPT_NO_STREAM", "SPT_REFERENCE_BVH", "SPT_PRIMARY_CHUNKS", "SPT_NO_LDS_TABLES", "SPT_STREAM_IFIF", "SPT_BEZ_DEFER"):
        os.environ.pop(k, None)
    os.environ.update(switches)
    sc = spt.load_scene(path)
    got = r.render_shard(sc, spt.OutputConfig(w, h))
    out = []
    for name, ref in refs.items():
        d = (got.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
        out.append("%s: %d px %s" % (name, int(d.sum()), np.argwhere(d)[:4].tolist()))
    print(switches, " | ".join(out), flush=True)
    sc.close()
