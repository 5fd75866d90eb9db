"""The `spt` command-line driver (flags of the reference's src/main.rs:26-41).  Without a GPU it must fail
loudly - the product has no CPU fallback; with one its PNG equals the Python binding's film."""
import os
import subprocess

import numpy as np
import pytest

import _util

spt = _util.load_pkg()
CLI = os.path.join(spt.LIB_DIR, "spt")
ARGS = ["-s", os.path.join(_util.SCENES, "cfg2_cube.json"), "-r", os.path.join(_util.SCENES, "pt.json"), "-w", "96", "-h", "64", "--spp", "16"]


def test_cli_usage_and_errors(tmp_path):
    assert subprocess.run([CLI], capture_output=True).returncode == 2
    assert subprocess.run([CLI, "--bogus"], capture_output=True).returncode == 2
    r = subprocess.run([CLI, "-s", str(tmp_path / "none.json"), "-r", ARGS[3], "-o", str(tmp_path / "o.png")], capture_output=True, text=True)
    assert r.returncode == 1 and "Error" in r.stderr


def test_cli_without_a_gpu_fails_loudly(tmp_path):
    if spt.device_count() > 0:
        pytest.skip("a GPU is visible")
    r = subprocess.run([CLI] + ARGS + ["-o", str(tmp_path / "o.png")], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr
    assert not (tmp_path / "o.png").exists()


@pytest.mark.gpu
def test_cli_png_equals_binding_film(tmp_path):
    out = tmp_path / "o.png"
    r = subprocess.run([CLI] + ARGS + ["-o", str(out), "--seed", "5"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = spt.read_png(out)[..., :3]
    sc = spt.load_scene(ARGS[1])
    ren = spt.load_renderer(ARGS[3], seed=5)
    ren.spp = 16
    film = ren.render_shard(sc, spt.OutputConfig(96, 64))
    assert np.array_equal(got, spt.film_to_rgb8(film))
    assert got.max() == 255 and (got == 85).any()      # the two lit cube faces
