"""Position-normal distributions ("glints": reference src/material/pndf_conductor.rs, src/bxdf/pndf_bvh.rs,
src/bxdf/microfacet.rs:56-170).

The reference holds no fixture for this material (its scenes/test_scene_15/16.json need scenes/textures/*.jpg, which are
not shipped), so the pins are independent float64 restatements in numpy, written from the formulas, not from
include/spt_pndf.h:
  * the Gaussian terms rebuilt from the normal map's texels,
  * brute-force sums over ALL terms against the tree walks (the walks cull boxes farther than 3 sigma, so a walk lies between
    the sum over the terms within reach and the sum over all of them),
  * the sampler against the density it claims.
Rendered pixels (GPU == oracle, bit for bit) are covered by tests/test_gpu_parity.py and the golden film.
"""
import json
import os

import numpy as np
import pytest

import _util

spt = _util.load_pkg()
SCENE = os.path.join(_util.SCENES, "t_pndf.json")
K = np.sqrt(8.0 * np.log(2.0))


@pytest.fixture(scope="module")
def scene():
    _util.ensure_cpu_build()
    return spt.load_scene(SCENE)


def tables(sc, k):
    pd = sc.array("pndfs")[k]
    terms = sc.array("pndf_terms")[int(pd["first_term"]):int(pd["first_term"]) + int(pd["n_terms"])]
    return pd, terms


def m2(cols):
    """glam Mat2 (column-major floats) -> (..., 2, 2) row-major matrices"""
    c = np.asarray(cols, dtype=np.float64)
    return np.stack([np.stack([c[..., 0], c[..., 2]], axis=-1), np.stack([c[..., 1], c[..., 3]], axis=-1)], axis=-2)


def normal_map(sc, image_index):
    im = sc.array("images")[image_index]
    lv = sc.array("image_levels")[int(im["first_level"])]
    w, h = int(lv["width"]), int(lv["height"])
    px = sc.array("texels")[int(lv["first_texel"]):int(lv["first_texel"]) + w * h]
    rgb = np.stack([(px >> s) & 255 for s in (0, 8, 16)], axis=-1).astype(np.float64) / 255.0
    return rgb.reshape(h, w, 3)


def bilinear(img, u, v):
    """sample_blinear (image_tex.rs:100-123) after the repeat wrap"""
    h, w, _ = img.shape
    u, v = u % 1.0, v % 1.0
    out = []
    for uu, vv in zip(u, v):
        x = uu * w
        x1 = int(np.floor(x + 0.5))
        x0 = x1 - 1
        xt = x - x0 - 0.5
        y = vv * h
        y1 = int(np.floor(y + 0.5))
        y0 = y1 - 1
        yt = y - y0 - 0.5
        x0, x1 = np.clip([x0, x1], 0, w - 1)
        y0, y1 = np.clip([y0, y1], 0, h - 1)
        c0 = img[y0, x0] * (1 - yt) + img[y1, x0] * yt
        c1 = img[y0, x1] * (1 - yt) + img[y1, x1] * yt
        out.append(c0 * (1 - xt) + c1 * xt)
    return np.array(out)


def normal_xy(img, u, v):
    c = bilinear(img, u, v) * 2.0 - 1.0
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    return c[:, :2]


@pytest.mark.parametrize("k,h,sigma_r", [(0, 1.0, 0.02), (1, 2.0, 0.05), (2, 1.5, 0.03)])
def test_terms_follow_the_normal_map(scene, k, h, sigma_r):
    pd, terms = tables(scene, k)
    img = [im for im in (normal_map(scene, q) for q in range(scene.desc.n_images)) if im.shape[:2] == (48, 64)][0]
    ih, iw, _ = img.shape
    nx, ny = int(iw / h), int(ih / h)
    assert int(pd["n_terms"]) == nx * ny
    hx, hy = 1.0 / nx, 1.0 / ny
    assert abs(pd["sigma_hx"] - hx / K) < 1e-7 and abs(pd["sigma_hy"] - hy / K) < 1e-7 and pd["sigma_r"] == np.float32(sigma_r)
    assert int(pd["s_block_count"]) == min(max(int(2.0 / (sigma_r * 16.0)), 1), 20)
    j, i = np.meshgrid(np.arange(nx), np.arange(ny))
    u, v = ((j + 0.5) * hx).ravel(), ((i + 0.5) * hy).ravel()
    assert np.allclose(terms["u"], np.stack([u, v], axis=1), atol=1e-6)
    inner = ((j > 0) & (j < nx - 1) & (i > 0) & (i < ny - 1)).ravel()   # (border cells sit on the wrap seam)
    s = normal_xy(img, u, v)
    assert np.abs(terms["s"] - s)[inner].max() < 2e-5
    dsdu = (normal_xy(img, u + 0.5 * hx, v) - normal_xy(img, u - 0.5 * hx, v)) * nx
    dsdv = (normal_xy(img, u, v + 0.5 * hy) - normal_xy(img, u, v - 0.5 * hy)) * ny
    jac = np.concatenate([dsdu, dsdv], axis=1)          # glam columns: (dsdu, dsdv)
    assert np.abs(terms["jacobian"] - jac)[inner].max() < 5e-3 * max(1.0, np.abs(jac).max())
    # PndfGaussTerm::new (pndf_bvh.rs:405-437) in float64 from the stored jacobian
    J = m2(terms["jacobian"])
    sh, sr = 1.0 / (float(pd["sigma_hx"]) * float(pd["sigma_hy"])), 1.0 / sigma_r ** 2
    A = sh * np.eye(2) + sr * np.swapaxes(J, 1, 2) @ J
    Ai = np.linalg.inv(A)
    B = sr * np.swapaxes(J, 1, 2)
    MU = Ai @ B
    S = sr * np.eye(2) - (sr * J) @ Ai @ B
    for name, want in (("mat_a", A), ("mat_mu", MU), ("mat_s", S)):
        got = m2(terms[name])
        assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max(), name


def test_trees_hold_every_term_once_and_boxes_enclose_them(scene):
    d = scene.desc
    nodes, refs, roots, terms = scene.array("pndf_nodes"), scene.array("pndf_refs"), scene.array("pndf_roots"), scene.array("pndf_terms")
    assert d.n_pndfs == 3 and len(refs) == 2 * len(terms)      # every term: once in its s-block list, once in the uv list

    def walk(root, first_ref, dims):
        seen = []
        todo = [int(root)]
        while todo:
            n = nodes[todo.pop()]
            lo, hi = n["bmin"], n["bmax"]
            ids = refs[first_ref + int(n["start"]):first_ref + int(n["end"])]
            pts = np.concatenate([terms["u"][ids], terms["s"][ids]], axis=1)
            if dims == 2:
                pts[:, 2:] = 0.0
            assert (pts >= lo - 1e-7).all() and (pts <= hi + 1e-7).all()
            if n["lc"] == 0xffffffff:
                assert 0 < len(ids) <= 5             # `end - start < max_leaf_size` or an unsplittable pair
                seen += ids.tolist()
            else:
                mid_l, mid_r = nodes[int(n["lc"])], nodes[int(n["rc"])]
                assert mid_l["start"] == n["start"] and mid_l["end"] == mid_r["start"] and mid_r["end"] == n["end"]
                todo += [int(n["lc"]), int(n["rc"])]
        return seen

    for pd in scene.array("pndfs"):
        first, n = int(pd["first_term"]), int(pd["n_terms"])
        sbc = int(pd["s_block_count"])
        seen = []
        for b in range(sbc * sbc):
            root, first_ref = roots[int(pd["first_root"]) + 2 * b], int(roots[int(pd["first_root"]) + 2 * b + 1])
            if root == 0xffffffff:
                continue
            ids = walk(root, first_ref, 4)
            s = terms["s"][ids]
            bx = np.minimum(((s[:, 0] + 1.0) * 0.5 * sbc).astype(int), sbc - 1)
            by = np.minimum(((s[:, 1] + 1.0) * 0.5 * sbc).astype(int), sbc - 1)
            assert (bx * sbc + by == b).all()
            seen += ids
        assert sorted(seen) == list(range(first, first + n))
        assert sorted(walk(pd["uv_root"], int(pd["uv_first_ref"]), 2)) == list(range(first, first + n))


def uv_values(pd, terms, u, sigma_p):
    sh2, sp2 = float(pd["sigma_hx"]) * float(pd["sigma_hy"]), sigma_p ** 2
    d2 = ((u[None, :] - terms["u"].astype(np.float64)) ** 2).sum(axis=1)
    return np.exp(-0.5 * d2 / (sh2 + sp2)) * sh2 / (sh2 + sp2)       # PndfUvBvh::find_terms (pndf_bvh.rs:350-366)


def within(pd, terms, u, sigma_p):
    lim = 3.0 * (np.array([pd["sigma_hx"], pd["sigma_hy"]], dtype=np.float64) + sigma_p)
    return (np.abs(u[None, :] - terms["u"]) <= lim[None, :]).all(axis=1)


@pytest.mark.parametrize("k", [0, 1, 2])
def test_footprint_sum_matches_brute_force(scene, k):
    pd, terms = tables(scene, k)
    rng = np.random.default_rng(5 + k)
    u = rng.random((48, 2)).astype(np.float32)
    lib = _util.oracle_lib()
    for sigma_p in (0.004, 0.03, 0.6):
        got = np.zeros(len(u), dtype=np.float32)
        lib.oracle_pndf_sum(scene.desc, k, sigma_p, len(u), u.ctypes.data, got.ctypes.data)
        for q in range(len(u)):
            val = uv_values(pd, terms, u[q].astype(np.float64), sigma_p)
            lo, hi = val[within(pd, terms, u[q].astype(np.float64), sigma_p)].sum(), val.sum()
            assert lo * (1 - 1e-4) <= got[q] <= hi * (1 + 1e-4), (sigma_p, q, lo, got[q], hi)
            if sigma_p > 0.5:                       # the reach covers the whole map: the walk IS the full sum
                assert abs(got[q] - hi) <= 1e-4 * hi
    assert got.min() > 0.0


def term_density(pd, terms, sigma_p, term_coe, u, s):
    """PndfGaussTerm::calc + integrate_gaussian_multiplication_2d (pndf_bvh.rs:447-466, 515-540), all terms at once"""
    sp_inv = 1.0 / sigma_p ** 2
    ds = s[None, :] - terms["s"].astype(np.float64)
    MU, S, A = m2(terms["mat_mu"]), m2(terms["mat_s"]), m2(terms["mat_a"])
    mu1 = terms["u"].astype(np.float64) + np.einsum("nij,nj->ni", MU, ds)
    c0 = 0.5 * sp_inv / np.pi
    c1 = term_coe * np.exp(-0.5 * np.einsum("ni,nij,nj->n", ds, S, ds))
    P0 = sp_inv * np.eye(2)
    Pi = P0[None] + A
    Sg = np.linalg.inv(Pi)
    mu = np.einsum("nij,nj->ni", Sg, (P0 @ u)[None, :] + np.einsum("nij,nj->ni", A, mu1))
    d0, d1 = mu - u[None, :], mu - mu1
    v0 = c0 * np.exp(-0.5 * np.einsum("ni,ij,nj->n", d0, P0, d0))
    v1 = c1 * np.exp(-0.5 * np.einsum("ni,nij,nj->n", d1, A, d1))
    return v0 * v1 * 2.0 * np.pi * np.sqrt(np.linalg.det(Sg))


@pytest.mark.parametrize("k", [0, 1, 2])
def test_density_matches_brute_force(scene, k):
    pd, terms = tables(scene, k)
    lib = _util.oracle_lib()
    rng = np.random.default_rng(17 + k)
    sbc = int(pd["s_block_count"])
    n = 40
    u = rng.random((n, 2)).astype(np.float32)
    # half vectors near the normals found at u (elsewhere the density is 0): the s of a nearby term plus a little noise
    near = np.array([np.argmin(((terms["u"] - uu) ** 2).sum(axis=1)) for uu in u])
    s = (terms["s"][near] + rng.normal(0.0, float(pd["sigma_r"]), (n, 2))).astype(np.float32)
    nonzero = 0
    for sigma_p in (0.006, 0.02):
        got = np.zeros(n, dtype=np.float32)
        lib.oracle_pndf_calc(scene.desc, k, sigma_p, n, u.ctypes.data, s.ctypes.data, got.ctypes.data)
        for q in range(n):
            uq, sq = u[q].astype(np.float64), s[q].astype(np.float64)
            val = uv_values(pd, terms, uq, sigma_p)
            inc = within(pd, terms, uq, sigma_p)
            # the walk's own sum decides the normalisation; bracket it like the test above
            dens = lambda coe: term_density(pd, terms, sigma_p, coe, uq, sq)
            base = dens(1.0 / (2.0 * np.pi * float(pd["sigma_r"]) ** 2))
            base = np.where(np.isfinite(base), base, 0.0)
            # PndfAccel::calc looks into ONE s-block only (pndf_bvh.rs:94-110)
            bx = min(max(int((sq[0] + 1.0) * 0.5 * sbc), 0), sbc - 1)
            by = min(max(int((sq[1] + 1.0) * 0.5 * sbc), 0), sbc - 1)
            tb = np.minimum(((terms["s"].astype(np.float64) + 1.0) * 0.5 * sbc).astype(int), sbc - 1)
            block = (tb[:, 0] == bx) & (tb[:, 1] == by)
            reach_s = (np.abs(sq[None, :] - terms["s"]) <= 3.0 * float(pd["sigma_r"])).all(axis=1)
            lo = base[block & inc & reach_s].sum() / val.sum()
            hi = base[block].sum() / val[inc].sum()
            assert lo * (1 - 2e-3) - 1e-5 <= got[q] <= hi * (1 + 2e-3) + 1e-5, (sigma_p, q, lo, got[q], hi)
            nonzero += got[q] > 1e-3
    assert nonzero > n            # more than half of the probes see glints


def test_sampler_draws_from_the_density(scene):
    lib = _util.oracle_lib()
    k, sigma_p = 0, 0.01
    pd, terms = tables(scene, k)
    u = np.array([0.37, 0.61], dtype=np.float32)
    n = 40000
    out = np.zeros((n, 4), dtype=np.float32)
    lib.oracle_pndf_sample_half(scene.desc, k, sigma_p, (spt.C.c_float * 2)(*u), 99, n, out.ctypes.data)
    h, pdf = out[:, :3], out[:, 3]
    assert np.allclose(np.linalg.norm(h, axis=1), 1.0, atol=1e-5) and (h[:, 2] >= 0).all()
    # the returned pdf is the density at the sampled s (microfacet.rs:136-139)
    sub = slice(0, 512)
    s = np.ascontiguousarray(h[sub, :2])
    uu = np.ascontiguousarray(np.repeat(u[None, :], 512, axis=0))
    dens = np.zeros(512, dtype=np.float32)
    lib.oracle_pndf_calc(scene.desc, k, sigma_p, 512, uu.ctypes.data, s.ctypes.data, dens.ctypes.data)
    ok = np.abs(h[sub, 2]) > 1e-3         # (normalising (sx, sy, sqrt(..)) moves s by an ulp; the density is smooth)
    assert np.allclose(dens[ok], pdf[sub][ok], rtol=2e-3, atol=1e-4)
    # histogram of the samples against the density integrated over the bins
    lo, hi = np.percentile(h[:, :2], 0.5, axis=0), np.percentile(h[:, :2], 99.5, axis=0)
    bins = 10
    hist, xe, ye = np.histogram2d(h[:, 0], h[:, 1], bins=bins, range=[[lo[0], hi[0]], [lo[1], hi[1]]])
    sub_n = 6
    gx = (xe[:-1, None] + (np.arange(sub_n) + 0.5)[None, :] * (xe[1] - xe[0]) / sub_n).ravel()
    gy = (ye[:-1, None] + (np.arange(sub_n) + 0.5)[None, :] * (ye[1] - ye[0]) / sub_n).ravel()
    G = np.stack(np.meshgrid(gx, gy, indexing="ij"), axis=-1).reshape(-1, 2).astype(np.float32)
    U = np.ascontiguousarray(np.repeat(u[None, :], len(G), axis=0))
    D = np.zeros(len(G), dtype=np.float32)
    lib.oracle_pndf_calc(scene.desc, k, sigma_p, len(G), U.ctypes.data, G.ctypes.data, D.ctypes.data)
    cell = (xe[1] - xe[0]) * (ye[1] - ye[0]) / sub_n ** 2
    mass = D.reshape(bins, sub_n, bins, sub_n).sum(axis=(1, 3)) * cell
    assert 0.5 < mass.sum() < 1.05          # a density over s (the one-block cut and the 3-sigma cull lose a little)
    freq = hist / n
    big = mass > 0.02
    assert big.sum() >= 3
    # the sampler draws from the un-cut mixture: where the density has its mass the two agree within the cut's loss
    assert np.abs(freq[big] - mass[big]).max() < 0.25 * mass[big].max()
    assert np.corrcoef(freq.ravel(), mass.ravel())[0, 1] > 0.9


def test_loader_contract(tmp_path, scene):
    rec = scene.array("material_recipes")
    glint = rec[rec["type"] >= 7]
    assert glint["type"].tolist() == [7, 7, 8] and glint["tex"][:, 1].tolist() == [0, 1, 2] and glint["ior"][2] == np.float32(1.5)
    mats = scene.array("materials")
    m = mats[mats["recipe"] != 0]
    # the constants are the fallback lobes: conductors with SPT_FRESNEL_SCHLICK (r0 = albedo), a Diffuse-substrate plastic
    assert m["fresnel"].tolist() == [1, 1, 0] and m["substrate"].tolist() == [0, 0, 1] and m["bxdf"].tolist() == [1, 2, 6]
    import shutil
    os.makedirs(tmp_path / "textures")
    shutil.copy(os.path.join(_util.SCENES, "textures", "scratch_normal.png"), tmp_path / "textures" / "scratch_normal.png")
    base = {"cameras": {"type": "perspective", "name": "c", "eye": [0.0, 0.0, 5.0], "forward": [0.0, 0.0, -1.0], "up": [0.0, 1.0, 0.0], "fov": 45.0},
            "textures": [{"type": "scalar", "name": "w", "value": [1.0, 0.5, 0.25]}, {"type": "scalar", "name": "r", "value": [0.2, 0.2, 0.2]},
                         {"type": "image", "name": "nm", "image_file": "textures/scratch_normal.png"}],
            "materials": [{"type": "pndf_conductor", "name": "g", "albedo": "w", "sigma_r": 0.03, "base_normal": "nm", "h": 4.0, "fallback_roughness": "r"}],
            "mediums": [], "surfaces": [], "primitives": [{"type": "sphere", "name": "s", "radius": 1.0}],
            "instances": [{"name": "i", "primitive": "s", "material": "g"}], "lights": []}
    (tmp_path / "a.json").write_text(json.dumps(base))
    sc = spt.load_scene(str(tmp_path / "a.json"))
    assert sc.desc.n_pndfs == 1 and sc.desc.n_pndf_terms == 16 * 12 and sc.array("pndfs")[0]["s_block_count"] == 4
    plastic = json.loads(json.dumps(base))
    plastic["materials"][0].update({"type": "pndf_plastic", "int_ior": 1.6, "ext_ior": 1.2})
    (tmp_path / "p.json").write_text(json.dumps(plastic))
    sp = spt.load_scene(str(tmp_path / "p.json"))
    assert sp.array("material_recipes")[0]["type"] == 8 and abs(sp.array("material_recipes")[0]["ior"] - 1.6 / 1.2) < 1e-6
    assert np.array_equal(sp.array("pndf_terms"), sc.array("pndf_terms"))
    for patch, status, word in (({"base_normal": "w"}, 102, "non-None dimensions"), ({"h": 0.0}, 102, "positive"), ({"sigma_r": -1.0}, 102, "positive"),
                                ({"h": 100.0}, 103, "P-NDF terms")):
        bad = json.loads(json.dumps(base))
        bad["materials"][0].update(patch)
        (tmp_path / "b.json").write_text(json.dumps(bad))
        with pytest.raises(spt.SptError) as e:
            spt.load_scene(str(tmp_path / "b.json"))
        assert e.value.status == status and word in str(e.value), str(e.value)


def test_oracle_renders_glints(scene):
    r = spt.PathTracer(max_depth=6, sampler=spt.SAMPLER_RANDOM, spp=8, seed=4)
    for cam in ("main", "graze"):
        film, st = _util.oracle_render(scene, r, 96, 72, camera=cam, flags=0)
        assert np.isfinite(film).all() and film.mean() > 0.1
    # the three glinting objects are lit (a broken normalisation shows as black or as overflow)
    film, _ = _util.oracle_render(scene, r, 96, 72, camera="main", flags=0)
    for name, (y0, y1, x0, x1) in (("conductor ball", (30, 50, 18, 38)), ("panel", (18, 28, 36, 60)), ("plastic ball", (46, 60, 42, 56))):
        box = film[y0:y1, x0:x1]
        assert 0.05 < box.mean() < 5.0, (name, box.mean())
