#!/usr/bin/env python3
"""Why does `k_shade<1>` run at 0.36 lane utilisation on the vertices of bounce >= 1 and at 0.69 on camera hits (cfg4)?
Variants of BASELINE configs[3] that remove one source of disagreement between the lanes of a wave at a time, rendered under a
`rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES` pass each (tools/shade_coherence_probe.sh):

  as_is        the scene as generated
  one_material every instance Lambert (the geometry, hence the rays, the queue order and the primitive kinds stay)
  no_delta     the smooth glass cube gets the rough glass material (every vertex runs the light-sample block)
  all_spheres  the cube and the floor plane become spheres of the same materials (one primitive kind in reconstruct_hit)

usage: python3 tools/shade_coherence_probe.py <variant>   (scenes are written to scenes_amd/generated/)"""
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scenes_amd"))
import bench  # noqa: E402
import make_scenes  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "as_is"
gen = make_scenes.make_full()
d = json.load(open(os.path.join(gen, "cfg4_materials_env.json")))
if variant == "one_material":
    for inst in d["instances"]:
        inst["material"] = "m_floor"
elif variant == "no_delta":
    for inst in d["instances"]:
        if inst["material"] == "m_glass":
            inst["material"] = "m_rough_glass"
elif variant == "all_spheres":
    for inst in d["instances"]:
        if inst["primitive"] == "cube":
            inst["primitive"] = "sphere"
            inst.pop("rotate", None)
        elif inst["primitive"] == "plane":
            inst.update({"primitive": "sphere", "scale": [60.0, 60.0, 60.0], "translate": [0.0, -61.0, 0.0]})
elif variant != "as_is":
    sys.exit(__doc__)
path = os.path.join(gen, "cfg4_probe_%s.json" % variant)
json.dump(d, open(path, "w"))
spt = bench.load_pkg()
sc = spt.load_scene(path)
r = spt.load_renderer(os.path.join(gen, "pt_random512.json"), seed=1)
r.spp = 128
cfg = spt.OutputConfig(1024, 1024, None, "main")
for _ in range(2):
    r.render_shard(sc, cfg, reuse_output=True)
st = r.last_stats
r.render_shard(sc, cfg, reuse_output=True, profile=True)
p = r.last_stats
print(variant, "ms %.2f" % st.gpu_ms, "vertices %.1f M" % (st.path_vertices / 1e6),
      {spt.KERNEL_NAMES[k]: round(p.kernel_ms[k], 2) for k in range(spt.N_KERNELS) if p.kernel_launches[k]})
