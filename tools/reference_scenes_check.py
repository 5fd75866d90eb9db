#!/usr/bin/env python3
"""Loads and renders the reference's own scenes/test_scene_*.json with this build, GPU against oracle.

The scene files stay in /root/reference (never committed) and the texture / HDR files they name are not shipped with
the reference at all, so this is a two-step developer tool, not a test:

  python tools/reference_scenes_check.py stage     (build container: copies scenes/ into the git-ignored
                                                    scenes_amd/generated/refscenes and writes stand-in PNG / JPEG /
                                                    EXR files for every missing `image_file` / `exr_file`)
  gpurun -- python tools/reference_scenes_check.py run     (GPU box: load, render 160 x 120 @ 8 spp, compare bits)
  python tools/reference_scenes_check.py clean

DESIGN.md ("The reference's own scene files") quotes its output."""
import glob
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.path.join(ROOT, "scenes_amd", "generated", "refscenes")
sys.path.insert(0, os.path.join(ROOT, "tests"))


def stage():
    import _util
    spt = _util.load_pkg()
    src = "/root/reference/scenes"
    if os.path.isdir(STAGE):
        shutil.rmtree(STAGE)
    shutil.copytree(src, STAGE)
    rng = np.random.default_rng(1)
    wanted = set()

    def walk(v):
        if isinstance(v, dict):
            for k, x in v.items():
                if k in ("image_file", "exr_file") and isinstance(x, str):
                    wanted.add(x)
                walk(x)
        elif isinstance(v, list):
            for x in v:
                walk(x)
    for p in glob.glob(os.path.join(STAGE, "*.json")):
        walk(json.load(open(p)))
    for rel in sorted(wanted):
        path = os.path.join(STAGE, rel)
        if os.path.exists(path):
            continue
        os.makedirs(os.path.dirname(path), exist_ok=True)
        y, x = np.mgrid[0:128, 0:128]
        img = np.stack([(x // 16 + y // 16) % 2 * 0.8 + 0.1, x / 128.0, y / 128.0], axis=-1).astype(np.float32)
        img = np.clip(img + rng.normal(0, 0.03, img.shape).astype(np.float32), 0, 1)
        if rel.endswith(".exr"):
            spt.write_exr(path, np.tile(img, (1, 2, 1)) * np.float32(2.0))
        elif rel.lower().endswith((".jpg", ".jpeg")):
            spt.write_jpeg(path, (img * 255).astype(np.uint8), 90)
        else:
            spt.write_png(path, img)
        print("stand-in", rel)


def run():
    import _util
    spt = _util.load_pkg()
    for p in sorted(glob.glob(os.path.join(STAGE, "test_scene_*.json"))):
        name = os.path.basename(p)
        try:
            sc = spt.load_scene(p)
        except spt.SptError as e:
            print(name, "load:", str(e)[:110].replace(STAGE, "."))
            continue
        r = spt.PathTracer(max_depth=8, sampler=spt.SAMPLER_RANDOM, spp=8, seed=5)
        w, h = 160, 120
        flags = _util.device_oracle_flags()
        if sc.desc.n_instances > 100:
            # the Catmull-Clark scenes (19 / 20: ~1 000 / ~250 patch instances): the exhaustive oracle clips EVERY patch for every
            # ray segment (hours of host time even for a thumbnail), so these two are compared in the other pairing - the
            # caller's trees on both sides (SPT_REFERENCE_BVH=1 vs the tree-walking oracle) - on a smaller image
            os.environ["SPT_REFERENCE_BVH"] = "1"
            flags = _util.ORACLE_DEVICE
            r.spp, w, h = 4, 96, 72
        print(name, "...", flush=True)
        ref, _ = _util.oracle_render(sc, r, w, h, flags=flags)
        got = r.render_shard(sc, spt.OutputConfig(w, h))
        os.environ.pop("SPT_REFERENCE_BVH", None)
        nan = np.isnan(ref)
        diff = int((got.view(np.uint32) != ref.view(np.uint32))[~nan].sum())
        print(name, "instances", sc.desc.n_instances, "triangles", sc.desc.n_tris, "mean %.4f" % float(np.nanmean(ref)),
              "NaN pixels", int(nan.any(axis=2).sum()), "same NaN", bool(np.array_equal(nan, np.isnan(got))), "words that differ", diff, flush=True)


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else ""
    if cmd == "stage":
        stage()
    elif cmd == "run":
        run()
    elif cmd == "clean":
        shutil.rmtree(STAGE, ignore_errors=True)
    else:
        sys.exit(__doc__)
