"""Where does a short timed region lose time?  Batches of K asynchronous steps of the headline workload, one process, back to back
(python tools/step_ramp.py [K] [batches]): ms per step of every batch."""
import os, sys, time, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spt = importlib.import_module("simple-path-tracer_amd")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
scene = spt.load_scene(os.path.join(ROOT, "scenes_amd", "cfg2_cube.json"))
r = spt.load_renderer(os.path.join(ROOT, "scenes_amd", "pt.json"), seed=1)
r.spp = 256
cfg = spt.OutputConfig(1024, 1024)
for _ in range(5):
    r.render_shard(scene, cfg, reuse_output=True)
out = []
for b in range(B):
    t0 = time.perf_counter()
    for _ in range(K):
        r.render_shard(scene, cfg, reuse_output=True, wait=False)
    t1 = time.perf_counter()
    r.wait(scene)
    t2 = time.perf_counter()
    out.append(((t2 - t0) * 1e3 / K, (t1 - t0) * 1e3 / K))
print("K =", K, " ms per step (queueing part):", "  ".join("%.3f (%.3f)" % o for o in out))
