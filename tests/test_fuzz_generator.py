"""The random-scene generator of tools/fuzz_scenes.py stays inside the schema: every scene it draws (both generator
variants) is accepted by the host loader and renders through the oracle without a crash.  (The GPU side of the fuzzer
is tests/test_gpu_fuzz.py and the campaigns in profiles/r01_fuzz_campaigns.txt.)"""
import importlib.util
import json
import os
import shutil

import numpy as np
import pytest

import _util


@pytest.mark.parametrize("v2", [False, True])
def test_generated_scenes_load_and_render(v2, monkeypatch):
    if v2:
        monkeypatch.setenv("FUZZ_V2", "1")
    else:
        monkeypatch.delenv("FUZZ_V2", raising=False)
    spec = importlib.util.spec_from_file_location("fuzz_scenes", os.path.join(_util.ROOT, "tools", "fuzz_scenes.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    spt = fz.spt
    work = fz.stage_assets()
    try:
        kinds, patches, media = set(), 0, 0
        for seed in range(40):
            rng = np.random.default_rng(500 + seed)
            scene = fz.make_scene(rng, work)
            path = os.path.join(work, "s.json")
            with open(path, "w") as fh:
                json.dump(scene, fh)
            sc = spt.load_scene(path)                      # raises on any schema slip of the generator
            kinds |= {m["type"] for m in scene["materials"]}
            patches += sc.desc.n_bezier_patches
            media += len(scene["mediums"])
            if seed < 6:
                r = spt.PathTracer(max_depth=3, sampler=spt.SAMPLER_RANDOM, spp=1, seed=seed)
                film, _ = _util.oracle_render(sc, r, 24, 18, threads=4)
                assert film.shape == (18, 24, 3)
            sc.close()
        assert kinds >= {"lambert", "conductor", "dielectric", "plastic", "pbr_metallic", "pbr_specular", "subsurface", "pseudo"}
        assert patches > 0 and media > 0
    finally:
        shutil.rmtree(work, ignore_errors=True)
