"""Subsurface substrate (src/bxdf/substrate.rs:182-350, src/material/subsurface.rs, pt.rs:147-151):
known answers for the diffusion profile, its radius table and the loader; the probe-ray path itself is covered by
the oracle-vs-GPU parity cases on scenes_amd/t_subsurface.json."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import _util

spt = _util.load_pkg()
lib = _util.oracle_lib()


def sp(d, r):
    out = (C.c_float * 3)()
    lib.oracle_ss_sp((C.c_float * 3)(*d), r, out)
    return np.array(out[:], dtype=np.float64)


def test_radius_table_is_the_cdf_of_the_profile():
    xy = (C.c_float * 2)()
    tab = []
    for i in range(512):
        lib.oracle_ss_cdf(i, xy)
        tab.append((xy[0], xy[1]))
    tab = np.array(tab, dtype=np.float64)
    i = np.arange(512)
    x = -2.0 * np.log(1.0 - i / 512.0)
    assert np.allclose(tab[:, 0], x, rtol=3e-6, atol=1e-7)
    assert np.allclose(tab[:, 1], 1.0 - np.exp(-x) * 0.25 - np.exp(-x / 3.0) * 0.75, atol=3e-7)
    assert (np.diff(tab[:, 1]) >= 0).all() and tab[0, 1] == 0.0           # the device bisects it
    # y(x) is the CDF of the normalised profile with d = 1: integral of Sp(r) 2 pi r dr from 0 to x
    for k in (40, 200, 400, 511):
        r = np.linspace(1e-6, tab[k, 0], 200_001)
        prof = (np.exp(-r) + np.exp(-r / 3.0)) / (8.0 * np.pi * r) * 2.0 * np.pi * r
        assert abs(np.trapezoid(prof, r) - tab[k, 1]) < 2e-5
    # sample_r inverts it piecewise linearly, -1 past the last entry
    assert abs(tab[-1, 1] - 0.98828) < 1e-5                               # the table stops short of 1: ~1.2 % of the draws fail
    for rand in (0.0, 0.013, 0.25, 0.5, 0.9, 0.988):
        xr = lib.oracle_ss_sample_r(rand)
        assert abs(np.interp(xr, tab[:, 0], tab[:, 1]) - rand) < 2e-6
    assert lib.oracle_ss_sample_r(float(tab[-1, 1]) + 1e-4) == -1.0


def test_diffusion_profile_is_normalised_per_channel():
    d = (0.07, 0.16, 0.4)
    r = np.geomspace(1e-7, 60.0, 400_001)
    vals = np.array([sp(d, float(x)) for x in r[::2000]])                 # spot values against the closed form
    for c in range(3):
        want = (np.exp(-r[::2000] / d[c]) + np.exp(-r[::2000] / (3.0 * d[c]))) / (8.0 * np.pi * d[c] * r[::2000])
        assert np.allclose(vals[:, c], want, rtol=2e-5)
        full = (np.exp(-r / d[c]) + np.exp(-r / (3.0 * d[c]))) / (8.0 * np.pi * d[c] * r)
        assert abs(np.trapezoid(full * 2.0 * np.pi * r, r) - 1.0) < 1e-4     # energy conserving


def test_loader_builds_the_substrate(tmp_path):
    sc = spt.load_scene(os.path.join(_util.SCENES, "t_subsurface.json"))
    mats = sc.array("materials")
    ss = mats[mats["substrate"] == 2]
    assert len(ss) == 3 and sorted(ss["bxdf"].tolist()) == [6, 6, 7]      # rough, image-backed rough, smooth
    rough = ss[(ss["bxdf"] == 6) & (ss["recipe"] == 0)][0]
    albedo, ld, ior = np.array([0.8, 0.55, 0.45], dtype=np.float32), np.float32(0.6), np.float32(1.4)
    assert np.allclose(rough["c0"], albedo) and abs(rough["ax"] - 0.09) < 1e-7 and rough["fresnel"] == 0
    assert np.allclose(rough["c1"], ld / (3.5 + 100.0 * (albedo - 0.33) ** 4), rtol=1e-6)     # Subsurface::new
    eta = 1.0 / 1.4                                                                     # Diffuse::new
    fm1 = 0.45966 - 1.73965 * eta + 3.37668 * eta**2 - 3.904945 * eta**3 + 2.49277 * eta**4 - 0.68441 * eta**5
    assert np.allclose(rough["c2"], albedo / np.pi / ((1 - albedo * 2 * fm1) * ior * ior), rtol=1e-5)
    rec = sc.array("material_recipes")
    assert len(rec) == 1 and rec[0]["type"] == 6 and abs(rec[0]["ior"] - 1.45) < 1e-7
    # missing key: the same message shape as the reference's get_str
    bad = json.load(open(os.path.join(_util.SCENES, "t_subsurface.json")))
    for m in bad["materials"]:
        if m["type"] == "subsurface":
            m.pop("ld")
    for t in bad["textures"]:
        if "image_file" in t:
            t["image_file"] = os.path.relpath(os.path.join(_util.SCENES, t["image_file"]), str(tmp_path))
    p = tmp_path / "bad.json"
    p.write_text(json.dumps(bad))
    with pytest.raises(spt.SptError) as e:
        spt.load_scene(str(p))
    assert "'ld'" in str(e.value)


def test_oracle_render_is_finite_and_translucent_objects_are_lit():
    sc = spt.load_scene(os.path.join(_util.SCENES, "t_subsurface.json"))
    r = spt.PathTracer(max_depth=8, sampler=spt.SAMPLER_RECURRENCE, spp=16, seed=4)
    film, _ = _util.oracle_render(sc, r, 96, 72, flags=_util.ORACLE_DEVICE)
    assert np.isfinite(film).all() and 0.1 < float(film.mean()) < 1.0
    assert film[30:50, 12:30].mean() > 0.05 and film[30:50, 66:84].mean() > 0.05      # the two spheres
