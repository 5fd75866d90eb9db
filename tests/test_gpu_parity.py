"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

The oracle runs in the configuration that defines the device's answer (tests/_util.device_oracle_flags): by
default the exhaustive one - no box culling at all, order-independent tie rule - because the library's own
padded trees never lose a hit; under SPT_REFERENCE_BVH=1 the ORACLE_DEVICE one (the caller's trees, reciprocal
slab test).  tests/test_oracle_pins.py bounds what these switches change.  Integer/index results must be bit-exact; radiance is
f32 and is expected bit-exact too (shared deterministic math, no FP contraction), with the
north-star tolerance (per-pixel mean L1 < 1e-3) as the hard gate.
"""
import os

import numpy as np
import pytest

import _util

pytestmark = pytest.mark.gpu

L1_TOL = 1e-3  # BASELINE.json north_star: per-pixel mean L1 < 1e-3


@pytest.fixture(scope="module")
def spt():
    return _util.load_pkg()


def _scene(spt, name):
    return spt.load_scene(os.path.join(_util.SCENES, name))


@pytest.mark.parametrize("scene_name", ["cfg1_sphere.json", "cfg2_cube.json", "t_materials.json", "t_power_is.json", "t_medium.json", "t_plastic.json", "t_textured.json", "t_gltf.gltf", "t_subsurface.json", "t_bezier.json", "t_pndf.json"])
def test_trace_closest_and_any_bit_exact(spt, scene_name):
    sc = _scene(spt, scene_name)
    rays = _util.random_rays(sc, 200_000, seed=11)
    ref = _util.oracle_trace_closest(sc, rays, _util.device_oracle_flags())
    got = sc.device_scene(0).trace_closest(rays)
    assert ref["instance"].max() >= 0, "test rays never hit"
    same_t = ref["t"].view(np.uint32) == got["t"].view(np.uint32)
    # Default mode (own trees vs the exhaustive oracle): every ray bit-identical, asserted below.
    # SPT_REFERENCE_BVH=1 (caller's trees vs the tree-walking oracle): the random rays also come from below the
    # floor, where objects resting on it give COINCIDENT surfaces; there box culling against an equal-depth
    # candidate depends on the visit order (the slab distance of a flat box and the triangle distance differ in
    # the last bit), hence the tolerance for the t_ scenes in that mode only.
    assert same_t.mean() > 0.997, same_t.mean()
    assert np.array_equal(ref["instance"] >= 0, got["instance"] >= 0)
    hit = ref["instance"] >= 0
    assert (np.abs(ref["t"][hit] - got["t"][hit]) <= 4e-7 * ref["t"][hit]).all()
    for f in ("instance", "prim"):
        assert np.array_equal(ref[f][same_t], got[f][same_t]), f
    for f in ("v", "w"):
        assert np.array_equal(ref[f][same_t].view(np.uint32), got[f][same_t].view(np.uint32)), f
    if not scene_name.startswith("t_") or not os.environ.get("SPT_REFERENCE_BVH"):
        assert same_t.all()      # own trees vs the exhaustive oracle: no box in either, nothing depends on visit order
    # any-hit with finite t_max taken around the closest hits
    rays2 = rays.copy()
    rays2["t_max"] = np.where(ref["instance"] >= 0, ref["t"] * np.float32(1.5), np.float32(5.0)).astype(np.float32)
    rays2["t_max"][::2] = (rays2["t_max"][::2] * np.float32(0.5)).astype(np.float32)
    occ_ref = _util.oracle_trace_any(sc, rays2, _util.device_oracle_flags())
    occ = sc.device_scene(0).trace_any(rays2)
    assert 0 < occ_ref.sum() < len(occ_ref)
    assert (occ_ref != occ).mean() < 1e-3
    if not scene_name.startswith("t_") or not os.environ.get("SPT_REFERENCE_BVH"):
        assert np.array_equal(occ_ref, occ)


@pytest.mark.parametrize("budget", [None, "0", "100000"])
@pytest.mark.parametrize("scene_name", ["cfg2_cube.json", "t_materials.json", "t_plastic.json"])
def test_exhaustive_walk_of_small_scenes(spt, scene_name, budget, monkeypatch):
    """flat.h (scenes of a handful of primitives: every lane tests every primitive; a mesh that few lanes of a wave reach is
    tested transposed, 64 / T rays x T triangles per pass) against the oracle and, through SPT_FLAT_BUDGET, against the
    tree walk: waves in which most rays miss the meshes (transposed), waves aimed at them (plain loops), a ragged last
    wave, rays with a negative t_min inside some waves (the plain loops: the transposed path's keys need t > 0), rays
    that start inside the objects, finite t_max."""
    monkeypatch.delenv("SPT_REFERENCE_BVH", raising=False)
    if budget is None:
        monkeypatch.delenv("SPT_FLAT_BUDGET", raising=False)
    else:
        monkeypatch.setenv("SPT_FLAT_BUDGET", budget)
    sc = _scene(spt, scene_name)
    flags = _util.device_oracle_flags()
    n = 64 * 700 + 37
    rays = _util.random_rays(sc, n, seed=29)
    rng = np.random.default_rng(31)
    inst = sc.array("instances")
    lo, hi = inst["bmin"].min(axis=0), inst["bmax"].max(axis=0)
    # first third: directions anywhere (most miss); second third: as random_rays aims them; last third: from inside the boxes
    k = n // 3
    d = rng.normal(size=(k, 3))
    rays["d"][:k] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["o"][2 * k:] = (lo + rng.uniform(0.05, 0.95, size=(n - 2 * k, 3)) * (hi - lo)).astype(np.float32)
    rays["t_min"][5::97] = np.float32(-2.0)        # a few per wave, some waves none
    rays["t_min"][64 * 300:64 * 301] = np.float32(0.0)
    ref = _util.oracle_trace_closest(sc, rays, flags)
    got = sc.device_scene(0).trace_closest(rays)
    assert 0.05 < (ref["instance"] >= 0).mean() < 0.95
    for f in ("t", "v", "w"):
        assert np.array_equal(ref[f].view(np.uint32), got[f].view(np.uint32)), f
    for f in ("instance", "prim"):
        assert np.array_equal(ref[f], got[f]), f
    rays2 = rays.copy()
    rays2["t_max"] = np.where(ref["instance"] >= 0, ref["t"] * np.float32(1.5), np.float32(5.0)).astype(np.float32)
    rays2["t_max"][::2] = (rays2["t_max"][::2] * np.float32(0.5)).astype(np.float32)
    occ_ref = _util.oracle_trace_any(sc, rays2, flags)
    assert 0 < occ_ref.sum() < n
    assert np.array_equal(occ_ref, sc.device_scene(0).trace_any(rays2))
    sc.close()


@pytest.mark.parametrize("aggregate", ["bvh", "group"])
def test_exhaustive_walk_with_awkward_mesh_sizes(spt, tmp_path, aggregate, monkeypatch):
    """flat.h on meshes whose triangle counts do not divide 64 (an icosahedron: 20, fans of 7, 5 and 3 triangles), one of them
    instanced twice under different transforms, spheres between them, BVH and GROUP aggregates: rays and films == oracle.
    SPT_FLAT_BUDGET lifts the scene (62 triangle tests per ray) into the exhaustive mode."""
    import json
    monkeypatch.delenv("SPT_REFERENCE_BVH", raising=False)
    monkeypatch.setenv("SPT_FLAT_BUDGET", "1000")
    os.makedirs(tmp_path / "models")
    g = (1.0 + 5.0 ** 0.5) / 2.0
    iv = [(-1, g, 0), (1, g, 0), (-1, -g, 0), (1, -g, 0), (0, -1, g), (0, 1, g), (0, -1, -g), (0, 1, -g), (g, 0, -1), (g, 0, 1), (-g, 0, -1), (-g, 0, 1)]
    it = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
          (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    def obj(name, verts, tris):
        with open(tmp_path / "models" / name, "w") as f:
            for v in verts:
                f.write("v %r %r %r\n" % tuple(float(c) for c in v))
            f.write("vn 0 1 0\nvt 0 0\n")
            for t in tris:
                f.write("f " + " ".join("%d/1/1" % (i + 1) for i in t) + "\n")
    def fan(n):
        ring = [(np.cos(2 * np.pi * k / (n + 1)), 0.2 * np.sin(3.0 * k), np.sin(2 * np.pi * k / (n + 1))) for k in range(n + 1)]
        return [(0.0, 0.3, 0.0)] + ring, [(0, 1 + k, 2 + k) for k in range(n)]
    obj("ico.obj", iv, it)
    for n in (7, 5, 3):
        v, t = fan(n)
        obj("fan%d.obj" % n, v, t)
    scene = {"cameras": {"type": "perspective", "name": "c", "eye": [0.0, 2.5, 9.0], "forward": [0.0, -0.2, -1.0], "up": [0.0, 1.0, 0.0], "fov": 45.0},
             "textures": [{"type": "scalar", "name": "w", "value": [0.8, 0.8, 0.75]}, {"type": "scalar", "name": "gn", "value": [0.14, 0.37, 1.44]},
                          {"type": "scalar", "name": "gk", "value": [3.98, 2.38, 1.6]}, {"type": "scalar", "name": "r", "value": [0.3, 0.3, 0.3]},
                          {"type": "scalar", "name": "one", "value": [1.0, 1.0, 1.0]}],
             "materials": [{"type": "lambert", "name": "m", "albedo": "w"}, {"type": "conductor", "name": "g", "ior": "gn", "ior_k": "gk", "roughness": "r"},
                           {"type": "dielectric", "name": "d", "int_ior": 1.5, "reflectance": "one", "transmittance": "one", "roughness": "r"}],
             "mediums": [], "surfaces": [],
             "primitives": [{"type": "trimesh", "name": "ico", "obj_file": "models/ico.obj"}, {"type": "trimesh", "name": "fan7", "obj_file": "models/fan7.obj"},
                            {"type": "trimesh", "name": "fan5", "obj_file": "models/fan5.obj"}, {"type": "trimesh", "name": "fan3", "obj_file": "models/fan3.obj"},
                            {"type": "sphere", "name": "ball", "radius": 1.0}],
             "instances": [{"name": "floor", "primitive": "fan7", "material": "m", "scale": [9.0, 1.0, 9.0], "translate": [0.0, -1.6, 0.0]},
                           {"name": "ico1", "primitive": "ico", "material": "g", "scale": [0.7, 0.7, 0.7], "translate": [-2.0, 0.0, 0.0]},
                           {"name": "ball1", "primitive": "ball", "material": "d", "scale": [0.8, 0.8, 0.8], "translate": [0.3, -0.4, 1.0]},
                           {"name": "ico2", "primitive": "ico", "material": "d", "scale": [0.5, 0.9, 0.5], "rotate": [20.0, 35.0, 0.0], "translate": [2.2, 0.2, -0.5]},
                           {"name": "f5", "primitive": "fan5", "material": "m", "scale": [1.5, 1.0, 1.5], "rotate": [60.0, 0.0, 10.0], "translate": [0.0, 1.8, -2.0]},
                           {"name": "f3", "primitive": "fan3", "material": "g", "scale": [1.2, 1.0, 1.2], "rotate": [-40.0, 20.0, 0.0], "translate": [-0.5, 0.9, 2.0]},
                           {"name": "ball2", "primitive": "ball", "material": "m", "scale": [0.4, 0.4, 0.4], "translate": [1.4, -1.0, 2.2]}],
             "lights": [{"type": "directional", "name": "sun", "direction": [-0.4, -1.0, -0.3], "strength": [3.0, 2.9, 2.7]}],
             "environment": {"type": "color", "color": [0.3, 0.35, 0.45]}}
    if aggregate == "group":
        scene["aggregate"] = "group"
    path = tmp_path / "awkward.json"
    path.write_text(json.dumps(scene))
    sc = spt.load_scene(str(path))
    assert sc.desc.n_instances == 7 and sc.desc.n_tris == 35
    flags = _util.device_oracle_flags()
    n = 64 * 900 + 11
    rays = _util.random_rays(sc, n, seed=41)
    d = np.random.default_rng(43).normal(size=(n // 2, 3))
    rays["d"][:n // 2] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)   # half of them anywhere: sparse box hits
    ref = _util.oracle_trace_closest(sc, rays, flags)
    got = sc.device_scene(0).trace_closest(rays)
    assert 0.1 < (ref["instance"] >= 0).mean() < 0.9 and len(set(ref["instance"].tolist())) == 8
    assert ref.tobytes() == got.tobytes()
    rays2 = rays.copy()
    rays2["t_max"] = np.where(ref["instance"] >= 0, ref["t"] * np.float32(1.5), np.float32(5.0)).astype(np.float32)
    rays2["t_max"][::2] = (rays2["t_max"][::2] * np.float32(0.5)).astype(np.float32)
    assert np.array_equal(_util.oracle_trace_any(sc, rays2, flags), sc.device_scene(0).trace_any(rays2))
    r = spt.PathTracer(max_depth=6, sampler=spt.SAMPLER_RANDOM, spp=8, seed=2)
    w, h = 160, 120
    ref_film, _ = _util.oracle_render(sc, r, w, h, flags=flags)
    film = r.render_shard(sc, spt.OutputConfig(w, h))
    nan = np.isnan(ref_film)
    assert nan.mean() < 1e-3 and np.array_equal(nan, np.isnan(film))
    assert np.array_equal(film.view(np.uint32)[~nan], ref_film.view(np.uint32)[~nan])
    sc.close()


@pytest.mark.parametrize("aggregate", ["bvh", "group"])
def test_eye_relative_copy_for_primary_rays(spt, tmp_path, aggregate, monkeypatch):
    """eye.h: camera rays of an LDS-resident scene whose meshes are used once each walk an eye-relative copy of the geometry
    (boxes as lo - o, triangles as s, s x e1, e2 . (s x e1), made on the host per camera position).  Three cameras one after the
    other on the same device scene (the copy is remade when the eye moves, reused when it does not), rotated / scaled instances,
    two spheres sharing one primitive, both aggregates: every film == the oracle's, and == the film without the copy."""
    import json
    monkeypatch.delenv("SPT_REFERENCE_BVH", raising=False)
    monkeypatch.delenv("SPT_NO_EYE_BLOB", raising=False)
    os.makedirs(tmp_path / "models")
    g = (1.0 + 5.0 ** 0.5) / 2.0
    iv = [(-1, g, 0), (1, g, 0), (-1, -g, 0), (1, -g, 0), (0, -1, g), (0, 1, g), (0, -1, -g), (0, 1, -g), (g, 0, -1), (g, 0, 1), (-g, 0, -1), (-g, 0, 1)]
    it = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
          (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    def obj(name, verts, tris):
        with open(tmp_path / "models" / name, "w") as f:
            for v in verts:
                f.write("v %r %r %r\n" % tuple(float(c) for c in v))
            f.write("vn 0 1 0\nvt 0 0\n")
            for t in tris:
                f.write("f " + " ".join("%d/1/1" % (i + 1) for i in t) + "\n")
    obj("ico.obj", iv, it)
    obj("quad.obj", [(-1, 0, -1), (1, 0, -1), (1, 0, 1), (-1, 0, 1)], [(0, 2, 1), (0, 3, 2)])
    obj("tri.obj", [(-1, 0, 0), (1, 0, 0), (0, 1.5, 0)], [(0, 1, 2)])
    cams = [{"type": "perspective", "name": "a", "eye": [0.0, 2.5, 9.0], "forward": [0.0, -0.2, -1.0], "up": [0.0, 1.0, 0.0], "fov": 45.0},
            {"type": "perspective", "name": "b", "eye": [6.5, 1.0, 2.0], "forward": [-1.0, -0.1, -0.3], "up": [0.0, 1.0, 0.0], "fov": 60.0},
            {"type": "perspective", "name": "c", "eye": [0.3, 0.4, 0.2], "forward": [0.2, -0.3, -1.0], "up": [0.0, 1.0, 0.0], "fov": 80.0}]   # between the objects
    scene = {"cameras": cams,
             "textures": [{"type": "scalar", "name": "w", "value": [0.8, 0.8, 0.75]}, {"type": "scalar", "name": "gn", "value": [0.14, 0.37, 1.44]},
                          {"type": "scalar", "name": "gk", "value": [3.98, 2.38, 1.6]}, {"type": "scalar", "name": "r", "value": [0.3, 0.3, 0.3]}],
             "materials": [{"type": "lambert", "name": "m", "albedo": "w"}, {"type": "conductor", "name": "g", "ior": "gn", "ior_k": "gk", "roughness": "r"}],
             "mediums": [], "surfaces": [],
             "primitives": [{"type": "trimesh", "name": "ico", "obj_file": "models/ico.obj"}, {"type": "trimesh", "name": "quad", "obj_file": "models/quad.obj"},
                            {"type": "trimesh", "name": "tri", "obj_file": "models/tri.obj"}, {"type": "sphere", "name": "ball", "radius": 1.0}],
             "instances": [{"name": "floor", "primitive": "quad", "material": "m", "scale": [9.0, 1.0, 9.0], "translate": [0.0, -1.6, 0.0]},
                           {"name": "ico1", "primitive": "ico", "material": "g", "scale": [0.5, 0.9, 0.5], "rotate": [20.0, 35.0, 10.0], "translate": [-1.8, 0.1, -0.5]},
                           {"name": "ball1", "primitive": "ball", "material": "m", "scale": [0.8, 0.8, 0.8], "translate": [1.3, -0.4, -1.0]},
                           {"name": "ball2", "primitive": "ball", "material": "g", "scale": [0.4, 0.7, 0.4], "rotate": [0.0, 0.0, 30.0], "translate": [2.4, -0.9, 1.2]},
                           {"name": "sail", "primitive": "tri", "material": "m", "scale": [1.5, 1.0, 1.0], "rotate": [-20.0, 40.0, 0.0], "translate": [0.2, -1.0, -2.5]}],
             "lights": [{"type": "directional", "name": "sun", "direction": [-0.4, -1.0, -0.3], "strength": [3.0, 2.9, 2.7]}],
             "environment": {"type": "color", "color": [0.3, 0.35, 0.45]}}
    if aggregate == "group":
        scene["aggregate"] = "group"
    path = tmp_path / "eye.json"
    path.write_text(json.dumps(scene))
    flags = _util.device_oracle_flags()
    r = spt.PathTracer(max_depth=5, sampler=spt.SAMPLER_RANDOM, spp=6, seed=3)
    w, h = 176, 132
    sc = spt.load_scene(str(path))
    films = {}
    for cam in ("a", "b", "c", "a", "b"):
        film = r.render_shard(sc, spt.OutputConfig(w, h, None, cam)).copy()
        if cam not in films:
            ref, _ = _util.oracle_render(sc, r, w, h, camera=cam, flags=flags)
            assert np.isfinite(ref).all() and ref.max() > 0.3
            films[cam] = ref
        assert np.array_equal(film.view(np.uint32), films[cam].view(np.uint32)), cam
    sc.close()
    monkeypatch.setenv("SPT_NO_EYE_BLOB", "1")
    sc = spt.load_scene(str(path))
    for cam in ("c", "a"):
        film = r.render_shard(sc, spt.OutputConfig(w, h, None, cam))
        assert np.array_equal(film.view(np.uint32), films[cam].view(np.uint32)), cam
    sc.close()


def test_trace_empty_batch(spt):
    sc = _scene(spt, "cfg2_cube.json")
    rays = np.zeros(0, dtype=spt.RAY_DTYPE)
    assert sc.device_scene(0).trace_closest(rays).shape == (0,)
    assert sc.device_scene(0).trace_any(rays).shape == (0,)


@pytest.mark.parametrize("scene_name,size,spp", [("cfg1_sphere.json", (96, 64), 16), ("cfg2_cube.json", (128, 128), 16)])
@pytest.mark.parametrize("sampler", ["recurrence", "random"])
def test_render_matches_oracle(spt, scene_name, size, spp, sampler):
    sc = _scene(spt, scene_name)
    r = spt.PathTracer(max_depth=8, sampler=spt.SAMPLER_RECURRENCE if sampler == "recurrence" else spt.SAMPLER_RANDOM,
                       spp=spp, seed=7)
    w, h = size
    ref, _ = _util.oracle_render(sc, r, w, h, flags=_util.device_oracle_flags())
    got = r.render_shard(sc, spt.OutputConfig(w, h), samples_per_pass=5)  # 16 = 5+5+5+1: exercises the pass loop
    l1 = float(np.abs(got - ref).mean())
    assert l1 < L1_TOL, l1
    assert ref.max() > 0.1
    mism = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
    assert mism == 0, "radiance not bit-exact: %d words differ, L1 %.3g" % (mism, l1)


@pytest.mark.parametrize("scene_name,camera,sampler", [
    ("t_materials.json", "main", "random"),      # GGX conductor (iso + aniso), mirror, rough + smooth glass, env MIS,
    ("t_materials.json", "top", "jittered"),     #   dir/point/spot/shape lights, uniform light sampler, TLAS
    ("t_power_is.json", "main", "recurrence"),   # power_is alias-table sampler, colour environment, group aggregate
    ("t_medium.json", None, "random"),           # homogeneous media (HG g=0.3 and isotropic), pseudo boundary, area light
    ("t_plastic.json", None, "random"),          # plastic (rough/smooth/aniso), pbr_metallic, pbr_specular
    ("t_textured.json", None, "random"),         # image textures: mips + trilinear (camera-ray differentials), wrap / tiling /
    ("t_textured.json", None, "recurrence"),     #   mode, sRGB, binary ops, normal + emissive maps, per-hit material recipes
    ("t_gltf.gltf", "cam", "random"),            # glTF import: metallic-roughness (G / B channels), spec-gloss (alpha), punctual lights
    ("t_subsurface.json", None, "random"),       # Subsurface substrate: BSSRDF probe rays, rough / smooth coat, image-backed albedo
    ("t_bezier.json", "main", "random"),         # bicubic Bezier patches (libspt_hip_bez.so): clipping, (u, v) texcoords, glass seen from
    ("t_bezier.json", "low", "recurrence"),      #   both sides, a medium boundary on a patch (its light samples probe the patch)
    ("t_catmull.json", "main", "random"),        # Catmull-Clark surfaces: 608 patch instances (regular + Gregory patches, creases, an open
    ("t_catmull.json", "side", "recurrence"),    #   tube) under the TLAS, far beyond LDS: the large-scene kernels of libspt_hip_bez.so
    ("t_pndf.json", "main", "random"),           # position-normal distributions (glints): conductor + plastic lobes, three tables, the GGX
    ("t_pndf.json", "graze", "recurrence"),      #   fallback on every bounce without a pixel footprint, Gaussian draws (Box-Muller)
])
def test_render_all_branches_match_oracle(spt, scene_name, camera, sampler):
    sc = _scene(spt, scene_name)
    kinds = {"random": spt.SAMPLER_RANDOM, "recurrence": spt.SAMPLER_RECURRENCE, "jittered": spt.SAMPLER_JITTERED}
    r = spt.PathTracer(max_depth=8, sampler=kinds[sampler], spp=16, division_x=4, division_y=4, seed=21)
    w, h = 160, 120
    # (t_catmull: testing every ray against 608 Bezier patches is out of reach, the tree-walking oracle stands in; it can
    #  lose a ray that grazes the edge of an exact box - none does in these two views)
    flags = _util.ORACLE_DEVICE if scene_name == "t_catmull.json" else _util.device_oracle_flags()
    ref, _ = _util.oracle_render(sc, r, w, h, camera=camera, flags=flags)
    got = r.render_shard(sc, spt.OutputConfig(w, h, None, camera), samples_per_pass=6)
    # texcoords at a sphere pole can be NaN in the reference too (acos of a normal.y a hair above 1,
    # sphere.rs:138-145): such pixels must be NaN on both sides, everything else bit-exact
    nan = np.isnan(ref)
    assert nan.mean() < 1e-3 and np.array_equal(nan, np.isnan(got))
    # (t_subsurface: a BSSRDF probe that lands where the profile underflows gives sp / pdf_pi = 0 / 0, pt.rs:150)
    if scene_name not in ("t_textured.json", "t_subsurface.json"):
        assert not nan.any()
    l1 = float(np.abs(got - ref)[~nan].mean())
    assert l1 < L1_TOL, l1
    mism = int((got.view(np.uint32) != ref.view(np.uint32))[~nan].sum())
    assert mism == 0, "radiance not bit-exact: %d words differ, L1 %.3g" % (mism, l1)


def test_tail_loop_kernel_gives_the_same_film_and_counters(spt, monkeypatch):
    """Fused scenes: once a pass has shown that few vertices are left after bounce 0, bounce 1 and everything behind it run
    in ONE launch whose lanes follow their paths to the end (k_shade's kLoop).  Same film, same segment counters."""
    monkeypatch.delenv("SPT_NO_TAIL_LOOP", raising=False)
    for name in ("cfg2_cube.json", "cfg1_sphere.json"):
        sc = _scene(spt, name)
        r = spt.PathTracer(max_depth=8, sampler=spt.SAMPLER_RANDOM, spp=24, seed=9)
        cfg = spt.OutputConfig(192, 160)
        ref, _ = _util.oracle_render(sc, r, 192, 160, flags=_util.device_oracle_flags())
        first = r.render_shard(sc, cfg).copy()                 # no hint yet: one launch per bounce
        st1 = r.last_stats
        one = (st1.segments_closest, st1.segments_shadow, st1.path_vertices, st1.vertices_second)
        second = r.render_shard(sc, cfg, profile=True).copy()  # the hint of the first render: the looping kernel
        st2 = r.last_stats
        assert st2.kernel_launches[1] == 1, "bounces >= 1 were not served by one launch"      # SPT_K_SHADE
        assert (st2.segments_closest, st2.segments_shadow, st2.path_vertices, st2.vertices_second) == one
        third = r.render_shard(sc, cfg, samples_per_pass=7)    # several passes
        for film in (first, second, third):
            assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
        monkeypatch.setenv("SPT_NO_TAIL_LOOP", "1")
        r.render_shard(sc, cfg, profile=True)
        assert r.last_stats.kernel_launches[1] == 7
        monkeypatch.delenv("SPT_NO_TAIL_LOOP")
        sc.close()


def test_shard_layout_does_not_change_pixels(spt):
    sc = _scene(spt, "cfg2_cube.json")
    r = spt.PathTracer(max_depth=8, spp=8, seed=3)
    w, h = 96, 80
    full = r.render_shard(sc, spt.OutputConfig(w, h))
    out = np.zeros_like(full)
    for k in range(3):
        rows = spt.shard_rows(h, k, 3, 16)
        out[rows] = r.render_shard(sc, spt.OutputConfig(w, h), shard_index=k, shard_count=3, strip_rows=16)
    assert np.array_equal(out.view(np.uint32), full.view(np.uint32))


def test_sample_chunks_of_the_primary_kernel_do_not_change_pixels(spt, monkeypatch):
    """k_primary<., kChunked>: any split of a pass's samples over workgroups gives the un-chunked film, also for a
    narrow shard (1 of 8) and with an environment (every miss writes its term into its own slot)."""
    for scene_name, cam in (("cfg2_cube.json", None), ("t_materials.json", "main")):
        sc = _scene(spt, scene_name)
        r = spt.PathTracer(max_depth=6, sampler=spt.SAMPLER_RANDOM, spp=24, seed=9)
        cfg = spt.OutputConfig(112, 96, None, cam)
        films = []
        for chunks in ("1", "3", "24", "64"):
            monkeypatch.setenv("SPT_PRIMARY_CHUNKS", chunks)
            films.append(r.render_shard(sc, cfg, samples_per_pass=24))
            films.append(r.render_shard(sc, cfg, shard_index=1, shard_count=8, strip_rows=4, samples_per_pass=10))
        for k in range(2, len(films)):
            assert np.array_equal(films[k].view(np.uint32), films[k % 2].view(np.uint32)), k


@pytest.mark.parametrize("bvh", ["own", "reference"])
@pytest.mark.parametrize("scene_name,camera", [("t_materials.json", "main"), ("t_medium.json", None), ("t_textured.json", None), ("t_gltf.gltf", "cam"),
                                               ("t_subsurface.json", None)])
def test_larger_multi_pass_renders_match_oracle(spt, scene_name, camera, bvh, monkeypatch):
    """3.5 M samples per scene in several passes (sample chunks, fused / un-fused bounces, refilling kernels as the
    scene selects them): still every word of the film equals the oracle's - the exhaustive oracle for the library's
    own trees, the tree-walking oracle for the caller's trees (the two oracles themselves differ in one pixel of
    t_gltf: a ray grazing the edge of an exact leaf box)."""
    if bvh == "reference":
        monkeypatch.setenv("SPT_REFERENCE_BVH", "1")
    else:
        monkeypatch.delenv("SPT_REFERENCE_BVH", raising=False)
    sc = _scene(spt, scene_name)
    r = spt.PathTracer(max_depth=8, sampler=spt.SAMPLER_RECURRENCE, spp=32, seed=77)
    w, h = 384, 288
    ref, _ = _util.oracle_render(sc, r, w, h, camera=camera, flags=_util.device_oracle_flags())
    got = r.render_shard(sc, spt.OutputConfig(w, h, None, camera), samples_per_pass=13)
    nan = np.isnan(ref)
    assert nan.mean() < 1e-4 and np.array_equal(nan, np.isnan(got))
    assert float(np.abs(got - ref)[~nan].mean()) < L1_TOL
    assert int((got.view(np.uint32) != ref.view(np.uint32))[~nan].sum()) == 0


def test_screen_space_bound_with_arbitrary_cameras(spt):
    """k_primary's pixel culling: cameras outside, at the edge of and INSIDE the scene's bounds, looking at, past and
    away from it, narrow and very wide, on a non-square image rendered as shards - always the oracle's film."""
    rng = np.random.default_rng(42)
    for scene_name in ("cfg2_cube.json", "t_materials.json"):
        sc = _scene(spt, scene_name)
        inst = sc.array("instances")
        lo, hi = inst["bmin"].min(axis=0), inst["bmax"].max(axis=0)
        centre, ext = (lo + hi) * 0.5, float((hi - lo).max())
        cams = []
        for k in range(10):
            radius = ext * (0.15, 0.6, 1.0, 2.5, 6.0)[k % 5]          # inside the bounds ... far away
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            eye = centre + d * radius
            aim = centre + rng.normal(size=3) * ext * (0.0, 0.3, 1.5)[k % 3]   # at the scene ... past it
            fwd = aim - eye if k != 7 else eye - centre                 # k = 7 looks away
            up = (0.0, 1.0, 0.0) if abs(fwd[1]) < 0.95 * np.linalg.norm(fwd) else (1.0, 0.0, 0.0)
            cams.append(spt.make_camera(eye, fwd, up, (20.0, 45.0, 90.0, 150.0)[k % 4]))
        r = spt.PathTracer(max_depth=5, sampler=spt.SAMPLER_RANDOM, spp=4, seed=13)
        w, h = 88, 56
        for k, cam in enumerate(cams):
            ref, _ = _util.oracle_render(sc, r, w, h, camera=cam, flags=_util.device_oracle_flags())
            got = np.zeros_like(ref)
            for s in range(3):
                rows = spt.shard_rows(h, s, 3, 8)
                got[rows] = r.render_shard(sc, spt.OutputConfig(w, h, None, cam), shard_index=s, shard_count=3, strip_rows=8)
            nan = np.isnan(ref)
            assert np.array_equal(nan, np.isnan(got)), (scene_name, k)
            assert np.array_equal(got.view(np.uint32)[~nan], ref.view(np.uint32)[~nan]), (scene_name, k)


def test_shards_written_in_place_into_a_full_film(spt):
    """out_strip_stride: three ranks' strips (the last one partial: 100 rows, strips of 16) DMA-ed straight into one
    film = the single-shard render; also through a page-locked shared-memory film."""
    sc = _scene(spt, "t_materials.json")
    r = spt.PathTracer(max_depth=5, sampler=spt.SAMPLER_RANDOM, spp=6, seed=8)
    cfg = spt.OutputConfig(72, 100, None, "main")
    full = r.render_shard(sc, cfg)
    film = np.full((100, 72, 3), -1.0, dtype=np.float32)
    for k in range(3):
        r.render_shard(sc, cfg, shard_index=k, shard_count=3, strip_rows=16, film=film)
    assert np.array_equal(film.view(np.uint32), full.view(np.uint32))
    shared = spt.SharedFilm(100, 72, create=True)
    try:
        shared.pin()
        shared.film[:] = -1.0
        for k in range(4):
            r.render_shard(sc, cfg, shard_index=k, shard_count=4, strip_rows=8, film=shared.film)
        assert np.array_equal(shared.film.view(np.uint32), full.view(np.uint32))
    finally:
        shared.close()


def test_many_instances_device_tlas(spt, tmp_path):
    """225 instances (cubes and spheres, random scales / rotations): the device-built TLAS (SAH over the instance
    boxes, leaves re-ordered behind an index table) returns the exhaustive oracle's hits and film."""
    import json
    import shutil
    os.makedirs(tmp_path / "models")
    shutil.copy(os.path.join(_util.SCENES, "models", "cube.obj"), tmp_path / "models" / "cube.obj")
    rng = np.random.default_rng(3)
    inst = [{"name": "floor", "primitive": "cube", "material": "m", "scale": [12.0, 0.1, 12.0], "translate": [0.0, -1.2, 0.0]}]
    for k in range(224):
        x, z = (k % 15 - 7) * 1.4, (k // 15 - 7) * 1.4
        sc = float(rng.uniform(0.15, 0.55))
        inst.append({"name": "i%03d" % k, "primitive": "cube" if k % 3 else "ball", "material": "m" if k % 2 else "g",
                     "scale": [sc, float(sc * rng.uniform(0.5, 2.0)), sc], "rotate": [float(rng.uniform(0, 90)), float(rng.uniform(0, 360)), 0.0],
                     "translate": [x + float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.8, 1.5)), z + float(rng.uniform(-0.3, 0.3))]})
    scene = {"cameras": {"type": "perspective", "name": "c", "eye": [0.0, 9.0, 16.0], "forward": [0.0, -0.5, -1.0], "up": [0.0, 1.0, 0.0], "fov": 40.0},
             "textures": [{"type": "scalar", "name": "w", "value": [0.8, 0.8, 0.75]}, {"type": "scalar", "name": "gold_n", "value": [0.14, 0.37, 1.44]},
                          {"type": "scalar", "name": "gold_k", "value": [3.98, 2.38, 1.6]}, {"type": "scalar", "name": "r", "value": [0.3, 0.3, 0.3]}],
             "materials": [{"type": "lambert", "name": "m", "albedo": "w"},
                           {"type": "conductor", "name": "g", "ior": "gold_n", "ior_k": "gold_k", "roughness": "r"}],
             "mediums": [], "surfaces": [],
             "primitives": [{"type": "trimesh", "name": "cube", "obj_file": "models/cube.obj"}, {"type": "sphere", "name": "ball", "radius": 1.0}],
             "instances": inst,
             "lights": [{"type": "directional", "name": "sun", "direction": [-0.4, -1.0, -0.3], "strength": [3.0, 2.9, 2.7]}],
             "environment": {"type": "color", "color": [0.3, 0.35, 0.45]}}
    path = tmp_path / "many.json"
    path.write_text(json.dumps(scene))
    sc = spt.load_scene(str(path))
    assert sc.desc.n_instances == 225
    rays = _util.random_rays(sc, 100_000, seed=9)
    ref = _util.oracle_trace_closest(sc, rays, _util.device_oracle_flags())
    got = sc.device_scene(0).trace_closest(rays)
    assert (ref["instance"] >= 0).mean() > 0.3
    assert ref.tobytes() == got.tobytes()
    r = spt.PathTracer(max_depth=6, sampler=spt.SAMPLER_RANDOM, spp=8, seed=2)
    w, h = 200, 150
    ref_film, _ = _util.oracle_render(sc, r, w, h, flags=_util.device_oracle_flags())
    film = r.render_shard(sc, spt.OutputConfig(w, h))
    nan = np.isnan(ref_film)
    assert nan.mean() < 1e-3 and np.array_equal(nan, np.isnan(film))
    assert np.array_equal(film.view(np.uint32)[~nan], ref_film.view(np.uint32)[~nan])


@pytest.mark.parametrize("scene_name,camera", [("t_subsurface.json", None), ("t_textured.json", None), ("t_materials.json", "main"), ("t_bezier.json", "low")])
def test_lds_geometry_without_lds_tables(spt, scene_name, camera, monkeypatch):
    """A scene whose traversal geometry fits LDS but whose shading tables do not runs the un-tabbed shade kernels; the
    BSSRDF probe inside k_shade<3> must then still walk the LDS node format (forced here with SPT_NO_LDS_TABLES)."""
    monkeypatch.setenv("SPT_NO_LDS_TABLES", "1")
    sc = _scene(spt, scene_name)
    r = spt.PathTracer(max_depth=6, sampler=spt.SAMPLER_RANDOM, spp=8, seed=17)
    w, h = 128, 96
    ref, _ = _util.oracle_render(sc, r, w, h, camera=camera, flags=_util.device_oracle_flags())
    got = r.render_shard(sc, spt.OutputConfig(w, h, None, camera), samples_per_pass=3)
    nan = np.isnan(ref)
    assert np.array_equal(nan, np.isnan(got))
    assert np.array_equal(got.view(np.uint32)[~nan], ref.view(np.uint32)[~nan])


def test_mesh_too_large_for_lds_matches_oracle(spt, tmp_path):
    """A smaller cousin of BASELINE config 5: a 24 k-triangle displaced sphere (deep BLAS: the traversal stack spills
    past its LDS levels; geometry far beyond LDS, so the global-memory kernels with 4-wide compressed nodes and the
    refilling walkers run without any switch), inside a box of haze, lit by an emissive quad only."""
    import importlib.util
    import json
    import shutil
    spec = importlib.util.spec_from_file_location("make_scenes", os.path.join(_util.SCENES, "make_scenes.py"))
    ms = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ms)
    n = ms.write_displaced_sphere(str(tmp_path / "blob.obj"), 160, 76, 11)
    assert n == 2 * 160 * 75
    for name in ("cube.obj", "plane.obj"):
        shutil.copy(os.path.join(_util.SCENES, "models", name), tmp_path / name)
    scene = {
        "cameras": {"type": "perspective", "name": "main", "eye": [0.0, 0.8, 6.0], "forward": [0.0, -0.1, -1.0], "up": [0.0, 1.0, 0.0], "fov": 40.0},
        "textures": [{"type": "scalar", "name": "white", "value": [0.8, 0.8, 0.8]}, {"type": "scalar", "name": "clay", "value": [0.7, 0.45, 0.3]}],
        "materials": [{"type": "lambert", "name": "m_floor", "albedo": "white"}, {"type": "lambert", "name": "m_clay", "albedo": "clay"},
                      {"type": "pseudo", "name": "m_pseudo"}],
        "mediums": [{"type": "homogeneous", "name": "haze", "sigma_a": [0.05, 0.05, 0.05], "asymmetric": 0.3}],
        "primitives": [{"type": "trimesh", "name": "blob", "obj_file": "blob.obj"}, {"type": "trimesh", "name": "cube", "obj_file": "cube.obj"},
                       {"type": "trimesh", "name": "plane", "obj_file": "plane.obj"}],
        "surfaces": [{"name": "s_haze", "material": "m_pseudo", "inside_medium": "haze"},
                     {"name": "s_light", "material": "m_floor", "emissive": [10.0, 10.0, 10.0]}],
        "instances": [{"name": "floor", "primitive": "plane", "material": "m_floor", "scale": [8.0, 1.0, 8.0], "translate": [0.0, -1.5, 0.0]},
                      {"name": "blob", "primitive": "blob", "material": "m_clay"},
                      {"name": "hazebox", "primitive": "cube", "surface": "s_haze", "scale": [2.0, 2.0, 2.0]},
                      {"name": "quad", "primitive": "plane", "surface": "s_light", "scale": [1.5, 1.0, 1.5], "rotate": [180.0, 0.0, 0.0], "translate": [0.0, 3.5, 0.0]}],
        "lights": [],
    }
    (tmp_path / "blob.json").write_text(json.dumps(scene))
    sc = spt.load_scene(str(tmp_path / "blob.json"))
    assert sc.desc.n_tris == n + 12 + 2
    rays = _util.random_rays(sc, 30_000, seed=4)
    ref = _util.oracle_trace_closest(sc, rays, _util.device_oracle_flags())
    got = sc.device_scene(0).trace_closest(rays)
    assert (ref["instance"] >= 0).mean() > 0.3
    if not os.environ.get("SPT_REFERENCE_BVH"):
        assert ref.tobytes() == got.tobytes()
    else:
        assert (ref["t"].view(np.uint32) == got["t"].view(np.uint32)).mean() > 0.999
    r = spt.PathTracer(max_depth=8, sampler=spt.SAMPLER_RECURRENCE, spp=4, seed=6)
    w, h = 128, 96
    ref_film, _ = _util.oracle_render(sc, r, w, h, flags=_util.device_oracle_flags())
    film = r.render_shard(sc, spt.OutputConfig(w, h))
    assert ref_film.max() > 0.05 and not np.isnan(ref_film).any()
    mism = int((film.view(np.uint32) != ref_film.view(np.uint32)).sum())
    assert mism == 0, mism


def test_mid_size_mesh_lds_geometry_but_global_tables(spt, tmp_path):
    """288 triangles: the traversal blob (positions, nodes) fits LDS, the shading tables (144 B of attributes per triangle)
    do not - the configuration SPT_NO_LDS_TABLES forces on the small scenes, here reached by the scene itself; with a
    Subsurface material so that the BSSRDF probe walks the LDS geometry from the un-tabbed kernel."""
    import importlib.util
    import json
    import shutil
    spec = importlib.util.spec_from_file_location("make_scenes", os.path.join(_util.SCENES, "make_scenes.py"))
    ms = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ms)
    assert ms.write_displaced_sphere(str(tmp_path / "blob.obj"), 24, 7, 5) == 288
    shutil.copy(os.path.join(_util.SCENES, "models", "plane.obj"), tmp_path / "plane.obj")
    scene = {
        "cameras": {"type": "perspective", "name": "main", "eye": [0.0, 1.0, 5.0], "forward": [0.0, -0.15, -1.0], "up": [0.0, 1.0, 0.0], "fov": 40.0},
        "textures": [{"type": "scalar", "name": "white", "value": [0.8, 0.8, 0.8]}, {"type": "scalar", "name": "skin", "value": [0.8, 0.55, 0.45]},
                     {"type": "scalar", "name": "ld", "value": [0.5, 0.5, 0.5]}, {"type": "scalar", "name": "r", "value": [0.3, 0.3, 0.3]}],
        "materials": [{"type": "lambert", "name": "m_floor", "albedo": "white"},
                      {"type": "subsurface", "name": "m_ss", "int_ior": 1.4, "albedo": "skin", "ld": "ld", "roughness": "r"}],
        "mediums": [],
        "primitives": [{"type": "trimesh", "name": "blob", "obj_file": "blob.obj"}, {"type": "trimesh", "name": "plane", "obj_file": "plane.obj"}],
        "surfaces": [{"name": "s_light", "material": "m_floor", "emissive": [9.0, 8.0, 7.0], "double_sided": True}],
        "instances": [{"name": "floor", "primitive": "plane", "material": "m_floor", "scale": [6.0, 1.0, 6.0], "translate": [0.0, -1.1, 0.0]},
                      {"name": "blob", "primitive": "blob", "material": "m_ss"},
                      {"name": "quad", "primitive": "plane", "surface": "s_light", "rotate": [180.0, 0.0, 0.0], "translate": [0.5, 3.0, 0.5]}],
        "lights": [{"type": "directional", "name": "sun", "direction": [-0.4, -1.0, -0.4], "strength": [1.5, 1.4, 1.3]}],
    }
    (tmp_path / "mid.json").write_text(json.dumps(scene))
    sc = spt.load_scene(str(tmp_path / "mid.json"))
    r = spt.PathTracer(max_depth=6, sampler=spt.SAMPLER_RANDOM, spp=8, seed=9)
    w, h = 128, 96
    ref, _ = _util.oracle_render(sc, r, w, h, flags=_util.device_oracle_flags())
    got = r.render_shard(sc, spt.OutputConfig(w, h), samples_per_pass=3)
    assert ref.max() > 0.1
    nan = np.isnan(ref)
    assert np.array_equal(nan, np.isnan(got))
    assert np.array_equal(got.view(np.uint32)[~nan], ref.view(np.uint32)[~nan])


def _same_film(got, ref):
    nan = np.isnan(ref)
    assert np.array_equal(nan, np.isnan(got))
    inf = np.isinf(ref)
    assert np.array_equal(ref[inf], got[inf])
    ok = ~nan
    assert np.array_equal(got[ok].view(np.uint32), ref[ok].view(np.uint32)), int((got.view(np.uint32) != ref.view(np.uint32))[ok].sum())


@pytest.mark.parametrize("scene_name,camera,sampler,radius", [
    ("cfg2_cube.json", None, "recurrence", 1.5),     # fused LDS pipeline, R = 1, every neighbour sample inside the box
    ("cfg2_cube.json", None, "random", 0.8),         # R = 1, the weight sum depends on the offsets
    ("cfg2_cube.json", None, "random", 0.3),         # R = 0: own samples only, counted when inside (empty boxes divide by 0)
    ("cfg2_cube.json", None, "jittered", 2.2),       # R = 2
    ("t_materials.json", "main", "random", 1.0),     # environment: every pixel is live; general shade kernel
    ("t_textured.json", None, "random", 1.2),        # textured shade kernel (its bounce 0 redoes the sampler draw)
])
def test_box_filter_of_any_radius_matches_oracle(spt, scene_name, camera, sampler, radius, monkeypatch):
    # film.rs:71-92 / boxf.rs: unweighted colours of (2R+1)^2 pixels over the count of in-radius samples
    sc = _scene(spt, scene_name)
    kinds = {"random": spt.SAMPLER_RANDOM, "recurrence": spt.SAMPLER_RECURRENCE, "jittered": spt.SAMPLER_JITTERED}
    r = spt.PathTracer(max_depth=6, sampler=kinds[sampler], spp=6, division_x=3, division_y=2, seed=13, filter_radius=radius)
    w, h = 112, 84
    ref, _ = _util.oracle_render(sc, r, w, h, camera=camera, flags=_util.device_oracle_flags())
    assert np.nanmax(ref[np.isfinite(ref)]) > 0.05
    got = r.render_shard(sc, spt.OutputConfig(w, h, None, camera), samples_per_pass=4)   # 6 = 4 + 2 samples per pass
    _same_film(got, ref)
    # bands of a few rows (each with its halo) instead of one band: same film
    monkeypatch.setenv("SPT_BOX_BAND_BYTES", str(w * 6 * 12 * 9))
    _same_film(r.render_shard(sc, spt.OutputConfig(w, h, None, camera)), ref)
    monkeypatch.delenv("SPT_BOX_BAND_BYTES")
    # shards: each rank traces the halo rows of its strips itself
    if radius > 0.5:
        out = np.zeros_like(ref)
        for k in range(3):
            rows = spt.shard_rows(h, k, 3, 8)
            out[rows] = r.render_shard(sc, spt.OutputConfig(w, h, None, camera), shard_index=k, shard_count=3, strip_rows=8)
            assert r.last_stats.samples > len(rows) * w * 6
        _same_film(out, ref)


def test_box_filter_negative_radius_is_the_reference_nan_film(spt):
    sc = _scene(spt, "cfg2_cube.json")
    r = spt.PathTracer(max_depth=2, spp=2, seed=1, filter_radius=-0.75)
    assert np.isnan(r.render_shard(sc, spt.OutputConfig(32, 16))).all()


def _tiny_scene(tmp_path, name, instances=True, lights=True, env=None, emissive=False):
    import json
    scene = {"cameras": {"type": "perspective", "name": "c", "eye": [0.0, 0.5, 4.0], "forward": [0.0, -0.1, -1.0], "up": [0.0, 1.0, 0.0], "fov": 45.0},
             "textures": [{"type": "scalar", "name": "w", "value": [0.7, 0.6, 0.5]}],
             "materials": [{"type": "lambert", "name": "m", "albedo": "w"}],
             "mediums": [],
             "surfaces": [{"name": "glow", "material": "m", "emissive": [2.0, 1.5, 1.0]}] if emissive else [],
             "primitives": [{"type": "sphere", "name": "ball", "radius": 1.0}],
             "instances": ([{"name": "a", "primitive": "ball", "material": "m"},
                            {"name": "b", "primitive": "ball", **({"surface": "glow"} if emissive else {"material": "m"}), "scale": [0.4, 0.4, 0.4], "translate": [1.6, 0.8, 0.0]}]
                           if instances else []),
             "lights": [{"type": "point", "name": "p", "position": [2.0, 3.0, 2.0], "strength": [20.0, 20.0, 20.0]}] if lights else []}
    if env is not None:
        scene["environment"] = {"type": "color", "color": env}
    path = tmp_path / (name + ".json")
    path.write_text(json.dumps(scene))
    return path


@pytest.mark.parametrize("kind", ["no_instances_env", "nothing_at_all", "no_lights", "no_lights_emissive", "env_only_light"])
def test_degenerate_scenes_match_oracle(spt, tmp_path, kind):
    """empty aggregates and empty light lists: defined results (DESIGN D5: the reference panics with zero lights)"""
    opts = {"no_instances_env": dict(instances=False, lights=True, env=[0.2, 0.4, 0.6]),
            "nothing_at_all": dict(instances=False, lights=False),
            "no_lights": dict(lights=False),
            "no_lights_emissive": dict(lights=False, emissive=True),
            "env_only_light": dict(lights=False, env=[0.5, 0.5, 0.5])}[kind]
    sc = spt.load_scene(str(_tiny_scene(tmp_path, kind, **opts)))
    r = spt.PathTracer(max_depth=4, sampler=spt.SAMPLER_RANDOM, spp=4, seed=3)
    w, h = 37, 23                                         # not a multiple of the 16 x 16 tiles
    ref, _ = _util.oracle_render(sc, r, w, h, flags=_util.device_oracle_flags())
    got = r.render_shard(sc, spt.OutputConfig(w, h))
    _same_film(got, ref)
    if kind == "no_instances_env":
        assert np.array_equal(np.unique(ref.reshape(-1, 3), axis=0), np.array([[0.2, 0.4, 0.6]], dtype=np.float32))
    if kind in ("nothing_at_all", "no_lights"):
        assert not ref.any()
    if kind in ("no_lights_emissive", "env_only_light"):
        assert ref.max() > 0.1
    rays = _util.random_rays(sc, 5000, seed=1)
    assert _util.oracle_trace_closest(sc, rays, _util.device_oracle_flags()).tobytes() == sc.device_scene(0).trace_closest(rays).tobytes()


@pytest.mark.parametrize("w,h,spp,depth", [(1, 1, 1, 1), (1, 7, 3, 0), (300, 1, 2, 8), (16, 16, 1, 255)])
def test_degenerate_render_sizes_match_oracle(spt, w, h, spp, depth):
    sc = _scene(spt, "cfg2_cube.json")
    r = spt.PathTracer(max_depth=depth, sampler=spt.SAMPLER_RECURRENCE, spp=spp, seed=11)
    ref, _ = _util.oracle_render(sc, r, w, h, flags=_util.device_oracle_flags())
    _same_film(r.render_shard(sc, spt.OutputConfig(w, h)), ref)
    if depth == 0:
        assert not ref.any()            # `while curr_depth < max_depth` (pt.rs:56) never runs


def test_more_shards_than_strips(spt):
    sc = _scene(spt, "cfg2_cube.json")
    r = spt.PathTracer(max_depth=3, spp=2, seed=1)
    w, h = 40, 20
    full = r.render_shard(sc, spt.OutputConfig(w, h))
    seen = np.zeros(h, dtype=bool)
    for k in range(5):                                    # 2 strips of 16 rows for 5 ranks: ranks 2 .. 4 own nothing
        rows = spt.shard_rows(h, k, 5, 16)
        part = r.render_shard(sc, spt.OutputConfig(w, h), shard_index=k, shard_count=5, strip_rows=16)
        assert part.shape == (len(rows), w, 3)
        if len(rows):
            assert np.array_equal(part.view(np.uint32), full[rows].view(np.uint32))
        seen[rows] = True
    assert seen.all()


def test_render_error_paths(spt):
    sc = _scene(spt, "cfg2_cube.json")
    r = spt.PathTracer(max_depth=8, spp=4)
    with pytest.raises(spt.SptError):
        r.render_shard(sc, spt.OutputConfig(0, 16))
    with pytest.raises(spt.SptError):
        r.render_shard(sc, spt.OutputConfig(16, 16), shard_index=2, shard_count=2)
    r2 = spt.PathTracer(max_depth=300, spp=4)
    with pytest.raises(spt.SptError):
        r2.render_shard(sc, spt.OutputConfig(16, 16))


def test_full_size_cfg2_properties(spt):
    """BASELINE configs[1] at full size (1024x1024 @ 256 spp, 268 M samples) through properties that do not
    need the oracle: closed-form face radiances, the reference's u8 values, determinism."""
    sc = _scene(spt, "cfg2_cube.json")
    r = spt.load_renderer(os.path.join(_util.SCENES, "pt.json"), seed=1)
    assert (r.spp, r.max_depth) == (256, 8)
    film = r.render_shard(sc, spt.OutputConfig(1024, 1024))
    assert r.last_stats.samples == 1024 * 1024 * 256
    l = np.array([1.0, 1.0, 1.0]) / np.sqrt(3.0)
    c, s = np.cos(np.radians(60.0)), np.sin(np.radians(60.0))
    lum = [5.0 / np.pi * max(float(np.dot(n, l)), 0.0) for n in ([s, 0.0, c], [-c, 0.0, s])]
    g = film[..., 0]
    assert np.array_equal(film[..., 0], film[..., 1]) and np.array_equal(film[..., 1], film[..., 2])
    near = lambda v: np.abs(g - v) < 3e-5
    interior = near(0.0) | near(lum[0]) | near(lum[1])
    assert interior.mean() > 0.99
    assert abs(near(lum[0]).mean() + near(lum[1]).mean() - 0.1846) < 0.004
    assert abs(float(g.mean()) - 0.11295) < 5e-4
    assert set(np.unique(spt.film_to_rgb8(film)[interior])) == {0, 85, 255}
    assert abs(r.last_stats.primary_hits / r.last_stats.samples - 0.1846) < 0.002
    # same call again: bit-identical (no float atomics, no order dependence); other pass size too
    again = r.render_shard(sc, spt.OutputConfig(1024, 1024), samples_per_pass=37)
    assert np.array_equal(film.view(np.uint32), again.view(np.uint32))


@pytest.mark.parametrize("scene_name,camera", [("cfg2_cube.json", None), ("t_materials.json", "main"), ("t_medium.json", None),
                                               ("t_plastic.json", None), ("t_subsurface.json", None), ("t_bezier.json", "main")])
def test_large_scene_path_matches_oracle(spt, scene_name, camera, monkeypatch):
    """The kernels a scene too big for LDS takes (global-memory geometry, compressed 4-wide BLAS nodes with
    conservatively widened child boxes, refilling shadow kernel), forced onto the small test scenes."""
    monkeypatch.setenv("SPT_NO_LDS_GEO", "1")
    sc = _scene(spt, scene_name)
    rays = _util.random_rays(sc, 100_000, seed=5)
    ref = _util.oracle_trace_closest(sc, rays, _util.device_oracle_flags())
    got = sc.device_scene(0).trace_closest(rays)
    assert np.array_equal(ref["instance"] >= 0, got["instance"] >= 0)
    same_t = ref["t"].view(np.uint32) == got["t"].view(np.uint32)
    assert same_t.mean() > 0.997, same_t.mean()
    for f in ("instance", "prim"):
        assert np.array_equal(ref[f][same_t], got[f][same_t]), f
    if not scene_name.startswith("t_"):
        assert same_t.all()
    r = spt.PathTracer(max_depth=8, sampler=spt.SAMPLER_RANDOM, spp=16, seed=33)
    w, h = 160, 120
    ref_film, _ = _util.oracle_render(sc, r, w, h, camera=camera, flags=_util.device_oracle_flags())
    got_film = r.render_shard(sc, spt.OutputConfig(w, h, None, camera), samples_per_pass=7)
    l1 = float(np.abs(got_film - ref_film).mean())
    assert l1 < L1_TOL, l1
    mism = int((got_film.view(np.uint32) != ref_film.view(np.uint32)).sum())
    assert mism == 0, "radiance not bit-exact: %d words differ, L1 %.3g" % (mism, l1)


@pytest.mark.parametrize("scene_name,camera", [("t_bezier.json", "main"), ("t_bezier.json", "low"), ("t_catmull.json", "main")])
def test_newton_iteration_build_of_the_patch_test_matches_oracle(spt, scene_name, camera, monkeypatch):
    """SPT_BEZIER_NI=1 (the reference's `bezier_ni` feature, bezier.rs:58-103; ABI v12: cp[0][0][3]): Newton's iteration
    instead of Bezier clipping, in the oracle and in every walker of libspt_hip_bez.so - rays and films bit for bit."""
    monkeypatch.setenv("SPT_BEZIER_NI", "1")
    sc = _scene(spt, scene_name)
    monkeypatch.delenv("SPT_BEZIER_NI")
    assert (sc.array("bezier_patches")["cp"][:, 0, 0, 3] == 1.0).all()
    flags = _util.ORACLE_DEVICE if scene_name == "t_catmull.json" else _util.device_oracle_flags()
    if scene_name != "t_catmull.json":
        rays = _util.random_rays(sc, 100_000, seed=5)
        ref = _util.oracle_trace_closest(sc, rays, flags)
        got = sc.device_scene(0).trace_closest(rays)
        assert (ref["instance"] >= 0).mean() > 0.2
        assert ref.tobytes() == got.tobytes()
        assert np.array_equal(_util.oracle_trace_any(sc, rays, flags), sc.device_scene(0).trace_any(rays))
    r = spt.PathTracer(max_depth=8, sampler=spt.SAMPLER_RANDOM, spp=12, seed=33)
    w, h = 160, 120
    ref_film, _ = _util.oracle_render(sc, r, w, h, camera=camera, flags=flags)
    for switches in ({}, {"SPT_STREAM_MASK": "0"}, {"SPT_BEZ_LDS": "1"}):
        for k, v in switches.items():
            monkeypatch.setenv(k, v)
        sc2 = _scene(spt, scene_name) if switches else sc
        if switches:
            monkeypatch.setenv("SPT_BEZIER_NI", "1")
            sc2.close()
            sc2 = _scene(spt, scene_name)
            monkeypatch.delenv("SPT_BEZIER_NI")
        got_film = r.render_shard(sc2, spt.OutputConfig(w, h, None, camera), samples_per_pass=5)
        for k in switches:
            monkeypatch.delenv(k)
        assert np.isfinite(ref_film).all() and ref_film.max() > 0.1
        mism = int((got_film.view(np.uint32) != ref_film.view(np.uint32)).sum())
        assert mism == 0, "%s: %d words differ" % (switches, mism)
        if sc2 is not sc:
            sc2.close()
    sc.close()
