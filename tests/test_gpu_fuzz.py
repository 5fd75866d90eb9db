"""A fixed handful of the random scenes of tools/fuzz_scenes.py (all material kinds, texture graphs, media, patches,
every light type, both aggregates / light samplers, random sampler / filter radius / pass size / shard layout):
GPU film == oracle film bit for bit, NaN positions included.  Seeds that ever failed a fuzz campaign stay in the list:
1164 - a BSSRDF exit point where sp / pdf_pi = 0 / 0 followed by an occluded light sample (the reference adds NaN * 0)."""
import importlib.util
import os
import shutil

import pytest

import _util

pytestmark = pytest.mark.gpu
SEEDS = [1164, 9, 24, 38, 47, 61, 3, 5, 14, 17, 20, 54, 60, 72]


@pytest.fixture(scope="module")
def fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_scenes", os.path.join(_util.ROOT, "tools", "fuzz_scenes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    work = mod.stage_assets()
    yield mod, work
    shutil.rmtree(work, ignore_errors=True)


@pytest.mark.parametrize("seed", SEEDS)
def test_random_scene_matches_oracle(fuzz, seed):
    mod, work = fuzz
    ok, info, _ = mod.run_seed(seed, work)
    assert ok, info


# FUZZ_V3 (round 2): the same generator with position-normal-distribution materials, Catmull-Clark surfaces, the new A/B
# switches (streaming / kind-sorted walkers, tail loop) drawn at random and frames queued asynchronously.  625: the
# tree-walking stand-in oracle (used for the film of scenes with hundreds of patches) loses a grazing hit there that the
# library keeps; the seed is settled against the exhaustive oracle.  94 / 131: Catmull-Clark scenes whose axis-aligned
# rays graze exact boxes.
SEEDS_V3 = [625, 94, 131, 2, 7, 11, 29, 33, 41, 248]


@pytest.mark.parametrize("seed", SEEDS_V3)
def test_random_scene_with_round2_features_matches_oracle(fuzz, seed, monkeypatch):
    mod, work = fuzz
    for k, v in (("FUZZ_V3", "1"), ("FUZZ_V2", "1"), ("FUZZ_SWITCHES", "1")):
        monkeypatch.setenv(k, v)
    try:
        ok, info, _ = mod.run_seed(seed, work)
    finally:
        for name in ("SPT_NO_FUSED", "SPT_NO_LDS_TABLES", "SPT_NO_LDS_GEO", "SPT_NO_PIXEL_CULL", "SPT_NO_OVERLAP", "SPT_NO_DYN_SHADOW", "SPT_NO_DYN_EXTEND",
                     "SPT_PRIMARY_CHUNKS", "SPT_BOX_BAND_BYTES", "SPT_BVH_MAX_LEAF", "SPT_DYN_BLOCKS", "SPT_NO_TAIL_LOOP", "SPT_NO_STREAM", "SPT_STREAM_MASK",
                     "SPT_STREAM_IFIF", "SPT_WST_MASK", "SPT_BEZ_LDS", "SPT_BEZ_DEFER"):
            os.environ.pop(name, None)       # run_seed sets the switches it drew in the process environment
    assert ok, info


# Larger images and more samples (FUZZ_SIZE_MUL=3, FUZZ_SPP_MUL=4) of two Catmull-Clark scenes that showed what the patch
# test's acceptance tolerance does to box culling (spt_hip.hip, bezier_box_margin): before the patch boxes were widened by
# that tolerance, the streaming walker's quantised boxes and the padded boxes of the other walkers saw different near misses
# (3034: two pixels), and in 3017 no walker agreed with either oracle.  Now every walker returns what testing every patch
# returns, and SPT_REFERENCE_BVH=1 what the reference's exact boxes return.
@pytest.mark.parametrize("seed", [3017, 3034])
def test_patch_tolerance_seeds(fuzz, seed, monkeypatch):
    mod, work = fuzz
    for k, v in (("FUZZ_V3", "1"), ("FUZZ_V2", "1"), ("FUZZ_SIZE_MUL", "3"), ("FUZZ_SPP_MUL", "4")):
        monkeypatch.setenv(k, v)
    monkeypatch.delenv("FUZZ_SWITCHES", raising=False)
    ok, info, _ = mod.run_seed(seed, work)
    assert ok, info
