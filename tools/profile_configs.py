#!/usr/bin/env python3
"""Per-kernel-class HIP-event times (SPT_RENDER_PROFILE) of cfg4 and of the per-GPU share of cfg5 at reduced spp:
where the time of the two larger BASELINE configs goes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402

spt = _util.load_pkg()
G = os.path.join(ROOT, "scenes_amd", "generated")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, scene, sampler, cam, w, h, shard_count in (("cfg4", "cfg4_materials_env.json", spt.SAMPLER_RANDOM, "main", 1024, 1024, 1),
                                                    ("cfg5 (1/8)", "cfg5_blob_medium.json", spt.SAMPLER_RECURRENCE, "main", 2048, 2048, 8)):
    sc = spt.load_scene(os.path.join(G, scene))
    r = spt.PathTracer(max_depth=8, sampler=sampler, spp=spp, seed=1)
    cfg = spt.OutputConfig(w, h, None, cam)
    r.render_shard(sc, cfg, shard_index=0, shard_count=shard_count, reuse_output=True)
    r.render_shard(sc, cfg, shard_index=0, shard_count=shard_count, reuse_output=True, profile=True)
    st = r.last_stats
    print("%s: %.1f M samples, %.2f ms (profiled, one stream) = %.1f Msamples/s; %.2f closest + %.2f shadow segments / sample" %
          (name, st.samples / 1e6, st.gpu_ms, st.samples / st.gpu_ms / 1e3, st.segments_closest / st.samples, st.segments_shadow / st.samples))
    for k, kn in enumerate(spt.KERNEL_NAMES):
        if st.kernel_launches[k]:
            print("    %-12s %8.3f ms  %3d launches" % (kn, st.kernel_ms[k], st.kernel_launches[k]))
    r.render_shard(sc, cfg, shard_index=0, shard_count=shard_count, reuse_output=True)
    print("    unprofiled: %.2f ms = %.1f Msamples/s" % (r.last_stats.gpu_ms, r.last_stats.samples / r.last_stats.gpu_ms / 1e3), flush=True)
