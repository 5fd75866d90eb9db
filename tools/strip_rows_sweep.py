import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util
spt = _util.load_pkg()
sc = spt.load_scene(os.path.join(ROOT, "scenes_amd", "cfg2_cube.json"))
r = spt.load_renderer(os.path.join(ROOT, "scenes_amd", "pt.json"), seed=1)
r.spp = 256
cfg = spt.OutputConfig(1024, 1024)
for world in (8, 4):
    for strip in (16, 8, 4, 2, 1):
        res = []
        for rank in range(world):
            for _ in range(3):
                r.render_shard(sc, cfg, shard_index=rank, shard_count=world, strip_rows=strip, reuse_output=True)
            t0 = time.perf_counter()
            n = 50
            for _ in range(n):
                r.render_shard(sc, cfg, shard_index=rank, shard_count=world, strip_rows=strip, reuse_output=True, wait=False)
            r.wait(sc)
            res.append((time.perf_counter() - t0) / n * 1e3)
        print("world %d strip_rows %2d: slowest %.3f fastest %.3f mean %.3f ms" % (world, strip, max(res), min(res), sum(res) / len(res)), flush=True)
