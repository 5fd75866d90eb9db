#!/usr/bin/env python3
"""cfg4 (BASELINE configs[3]) at 1024^2 @ 64 spp: how many path vertices sample a light (push a shadow ray), per bounce class.
Measured: 0.63 at bounce 0, 0.68 later - i.e. ~30 % of the vertices are delta lobes or black, the most a delta / non-delta
split of the shade queue could take out of the light-sample block.  GPU box: gpurun -- python3 tools/cfg4_vertex_stats.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT") or "/root/repo"
sys.path.insert(0, ROOT)
import bench
spt = bench.load_pkg()
import json
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scenes_amd"))
import make_scenes
if not os.path.exists(os.path.join(ROOT, "scenes_amd/generated/cfg4_materials_env.json")):
    make_scenes.main(); make_scenes.make_full()
sc = spt.load_scene(os.path.join(ROOT, "scenes_amd/generated/cfg4_materials_env.json"))
r = spt.load_renderer(os.path.join(ROOT, "scenes_amd/generated/pt_random512.json"), seed=1)
r.spp = 64
r.render_shard(sc, spt.OutputConfig(1024, 1024, None, "main"), reuse_output=True)
st = r.last_stats
print("samples", st.samples, "primary_hits", st.primary_hits, "path_vertices", st.path_vertices, "shadow_first", st.shadow_first, "seg_shadow", st.segments_shadow, "vertices_second", st.vertices_second)
later_v = st.path_vertices - st.primary_hits
later_s = st.segments_shadow - st.shadow_first
print("bounce 0: shadow / vertex = %.3f; later bounces: %.3f (vertices %d)" % (st.shadow_first / st.primary_hits, later_s / later_v, later_v))
