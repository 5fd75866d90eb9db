#!/usr/bin/env python3
"""Random scenes, GPU against oracle, bit for bit.

Every seed draws a scene from the whole supported schema - all material kinds (constant and image-textured
parameters), texture graphs, normal / emissive maps, media, spheres / meshes / Bezier patches with random transforms,
every light type, colour / EXR / no environment, both aggregates and light samplers - and a renderer configuration
(sampler, spp, depth, box-filter radius, samples per pass, shard layout), renders it through the C ABI and through
the oracle and compares the films word by word (NaN positions must agree); it also sends 6 000 adversarial rays per
scene (axis-aligned directions, origins exactly on a surface, odd t ranges) through the closest- / any-hit seams.
A developer tool, not a test:

    gpurun -- python tools/fuzz_scenes.py --seeds 0:40          (GPU box; exit code 1 and the seed on any mismatch)

The scene JSON of a failing seed is left in gpurun_out/fuzz/; `--seeds N:N+1` reproduces it.  tests/test_gpu_fuzz.py
runs a fixed handful of seeds (including every seed that ever failed) with the GPU suite."""
import argparse
import json
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _util  # noqa: E402

spt = _util.load_pkg()
SCENES = os.path.join(ROOT, "scenes_amd")


def f3(rng, lo, hi):
    return [float(x) for x in rng.uniform(lo, hi, 3)]


_make_scenes = None


def _blob(rng, work):
    """FUZZ_V2: a displaced sphere of a random resolution per seed (120 .. 19 000 triangles): geometry and shading tables
    land on either side of the LDS thresholds by themselves, deep trees spill the traversal stack"""
    global _make_scenes
    if _make_scenes is None:
        import importlib.util
        spec = importlib.util.spec_from_file_location("make_scenes", os.path.join(SCENES, "make_scenes.py"))
        _make_scenes = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_make_scenes)
    nu, nv = [(12, 6), (24, 7), (40, 12), (64, 20), (100, 40), (160, 60)][int(rng.integers(0, 6))]
    _make_scenes.write_displaced_sphere(os.path.join(work, "models", "blob_fuzz.obj"), nu, nv, int(rng.integers(0, 1000)))
    return "models/blob_fuzz.obj"


def make_scene(rng, work=None):
    tex = [{"type": "scalar", "name": "c%d" % k, "value": f3(rng, 0.05, 0.95)} for k in range(4)]
    tex += [{"type": "scalar", "name": "r%d" % k, "value": [float(v)] * 3} for k, v in enumerate((0.0, 0.12, 0.3, 0.55))]
    tex += [{"type": "scalar", "name": "eta", "value": f3(rng, 0.1, 1.6)}, {"type": "scalar", "name": "kk", "value": f3(rng, 1.5, 4.0)},
            {"type": "scalar", "name": "one", "value": [1.0, 1.0, 1.0]}, {"type": "scalar", "name": "ld", "value": f3(rng, 0.2, 0.8)}]
    images = ["checker.png", "noise_rgba.png", "rough_ramp.png", "stripes_ga.png"]
    wraps = ["repeat", "clamp", "mirror_repeat", "mirror_clamp"]
    img_names = []
    for k in range(int(rng.integers(0, 4))):
        t = {"type": "image", "name": "img%d" % k, "image_file": "textures/" + images[int(rng.integers(0, len(images)))]}
        if rng.random() < 0.6:
            t["tiling"] = [float(rng.uniform(0.5, 5.0)), float(rng.uniform(0.5, 5.0))]
        if rng.random() < 0.3:
            t["offset"] = [float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))]
        if rng.random() < 0.5:
            t["wrap"] = wraps[int(rng.integers(0, 4))]
        if rng.random() < 0.3:
            t["is_srgb"] = True
        if rng.random() < 0.15:
            t["mode"] = ["position", "normal", "tangent", "bitangent"][int(rng.integers(0, 4 if os.environ.get("FUZZ_V2") else 2))]
            t["tiling"] = f3(rng, 0.3, 1.5)
            t.pop("offset", None)
        tex.append(t)
        img_names.append(t["name"])
    if img_names and rng.random() < 0.5:
        tex.append({"type": ["mul", "add", "sub"][int(rng.integers(0, 3))], "name": "op0", "t1": img_names[0], "t2": "c0"})
        img_names.append("op0")
        if os.environ.get("FUZZ_V2") and rng.random() < 0.5:     # a second level: (img op c0) / one, sRGB-decoded
            tex.append({"type": "div", "name": "op1", "t1": "op0", "t2": "one", "is_srgb": bool(rng.random() < 0.5)})
            img_names.append("op1")
    normal_map = None
    if rng.random() < 0.3:
        tex.append({"type": "image", "name": "bumps", "image_file": "textures/bumps_normal.png", "tiling": [float(rng.uniform(1, 5))] * 2})
        normal_map = "bumps"

    def colour():
        return img_names[int(rng.integers(0, len(img_names)))] if img_names and rng.random() < 0.4 else "c%d" % int(rng.integers(0, 4))

    def rough():
        if img_names and rng.random() < 0.2:
            return "img0" if "img0" in img_names else "r2"
        return "r%d" % int(rng.integers(0, 4))

    def roughness_fields(m):
        if rng.random() < 0.3:
            m["roughness_x"], m["roughness_y"] = rough(), rough()
        else:
            m["roughness"] = rough()
        return m

    mats = []
    kinds = ["lambert", "conductor", "dielectric", "plastic", "pbr_metallic", "pbr_specular", "subsurface", "lambert", "conductor"]
    for k in range(int(rng.integers(2, 7))):
        kind = kinds[int(rng.integers(0, len(kinds)))]
        m = {"type": kind, "name": "m%d" % k}
        if kind == "lambert":
            m["albedo"] = colour()
        elif kind == "conductor":
            m.update(ior="eta", ior_k="kk")
            roughness_fields(m)
        elif kind == "dielectric":
            m.update(int_ior=float(rng.uniform(1.2, 1.8)), reflectance="one", transmittance=colour() if rng.random() < 0.3 else "one")
            if rng.random() < 0.5:
                m["ext_ior"] = 1.0
            roughness_fields(m)
        elif kind == "plastic":
            m.update(int_ior=float(rng.uniform(1.3, 1.7)), albedo=colour())
            if os.environ.get("FUZZ_V2") and rng.random() < 0.4:
                m["ext_ior"] = float(rng.uniform(1.0, 1.2))
            roughness_fields(m)
        elif kind == "pbr_metallic":
            m.update(base_color=colour(), metallic="c%d" % int(rng.integers(0, 4)))
            roughness_fields(m)
        elif kind == "pbr_specular":
            m.update(diffuse=colour(), specular="c%d" % int(rng.integers(0, 4)))
            roughness_fields(m)
        else:
            m.update(int_ior=float(rng.uniform(1.3, 1.6)), albedo=colour(), ld="ld")
            roughness_fields(m)
        mats.append(m)
    if os.environ.get("FUZZ_V3") and rng.random() < 0.45:
        # FUZZ_V3 (round 2): position-normal-distribution materials over the scratched normal map
        tex.append({"type": "image", "name": "scratch", "image_file": "textures/scratch_normal.png",
                    "tiling": [float(rng.uniform(1, 6)), float(rng.uniform(1, 6))], "offset": [float(rng.uniform(0, 1)), float(rng.uniform(0, 1))]})
        for k in range(int(rng.integers(1, 3))):
            m = {"type": "pndf_conductor" if rng.random() < 0.6 else "pndf_plastic", "name": "glint%d" % k, "albedo": colour(),
                 "sigma_r": float(rng.choice([0.01, 0.02, 0.05, 0.08])), "base_normal": "scratch", "h": float(rng.choice([1.0, 2.0, 3.0])),
                 "fallback_roughness": "r%d" % int(rng.integers(0, 4))}
            if m["type"] == "pndf_plastic":
                m["int_ior"] = float(rng.uniform(1.3, 1.7))
            mats.append(m)
    mats.append({"type": "pseudo", "name": "pseudo"})
    mediums = []
    for k in range(int(rng.integers(0, 3))):
        md = {"type": "homogeneous", "name": "med%d" % k, "sigma_a": f3(rng, 0.02, 0.8), "asymmetric": float(rng.choice([0.0, 0.3, -0.4, 0.005]))}
        if rng.random() < 0.5:
            md["sigma_s"] = f3(rng, 0.1, 4.0)      # read but unused by the reference (quirk Q4): must not matter
        mediums.append(md)
    prims = [{"type": "sphere", "name": "ball", "radius": 1.0},
             {"type": "sphere", "name": "off_ball", "center": f3(rng, -0.3, 0.3), "radius": float(rng.uniform(0.3, 0.8))},
             {"type": "trimesh", "name": "cube", "obj_file": "models/cube.obj"},
             {"type": "trimesh", "name": "plane", "obj_file": "models/plane.obj"},
             {"type": "trimesh", "name": "blob", "obj_file": _blob(rng, work) if (work and os.environ.get("FUZZ_V2")) else "models/blob_small.obj"}]
    use_patches = rng.random() < 0.35
    if use_patches:
        h = rng.uniform(-0.5, 1.2, (4, 4))
        prims.append({"type": "cubic_bezier", "name": "patch",
                      "control_points": [[[float(-1.5 + j + rng.uniform(-0.2, 0.2)), float(h[i][j]), float(1.5 - i + rng.uniform(-0.2, 0.2))] for j in range(4)] for i in range(4)]})
    use_cc = bool(os.environ.get("FUZZ_V3")) and rng.random() < 0.3
    if use_cc:       # Catmull-Clark surfaces: every patch an instance under the TLAS
        model = ["cc_cube.ply", "cc_cube_crease.ply", "cc_lshape.ply", "cc_tube.ply"][int(rng.integers(0, 4))]
        prims.append({"type": "catmull_clark", "name": "cc", "ply_file": "models/" + model, "fas_times": int(rng.integers(1, 3))})
    surfaces = []
    for k in range(int(rng.integers(0, 4))):
        s = {"name": "s%d" % k, "material": mats[int(rng.integers(0, len(mats)))]["name"]}
        roll = rng.random()
        if roll < 0.45:
            s["emissive"] = f3(rng, 1.0, 12.0)
            if img_names and rng.random() < 0.3:
                s["emissive_map"] = img_names[0]
        elif roll < 0.75 and mediums:
            s["material"] = "pseudo" if rng.random() < 0.7 else s["material"]
            s["inside_medium"] = mediums[int(rng.integers(0, len(mediums)))]["name"]
        if rng.random() < 0.3:
            s["double_sided"] = True
        if normal_map and rng.random() < 0.5:
            s["normal_map"] = normal_map
        surfaces.append(s)
    inst = [{"name": "floor", "primitive": "plane", "material": mats[0]["name"], "scale": [6.0, 1.0, 6.0], "translate": [0.0, -1.2, 0.0]}]
    names = ["ball", "off_ball", "cube", "plane", "blob"] + (["patch", "patch"] if use_patches else []) + (["cc"] if use_cc else [])
    for k in range(int(rng.integers(2, 9))):
        prim = names[int(rng.integers(0, len(names)))]
        i = {"name": "i%d" % k, "primitive": prim}
        s = None
        if surfaces and rng.random() < 0.5:
            s = surfaces[int(rng.integers(0, len(surfaces)))]
            if prim in ("patch", "cc") and "emissive" in s:
                s = None                                    # CubicBezier cannot be a shape light (bezier.rs:188-190)
        if s is not None:
            i["surface"] = s["name"]
        else:
            i["material"] = mats[int(rng.integers(0, len(mats) - 1))]["name"]
        sc = float(rng.uniform(0.3, 1.1))
        i["scale"] = [sc, sc * float(rng.uniform(0.6, 1.5)), sc] if rng.random() < 0.5 else [sc, sc, sc]
        if rng.random() < 0.6:
            i["rotate"] = f3(rng, -180.0, 180.0)
        i["translate"] = [float(rng.uniform(-2.5, 2.5)), float(rng.uniform(-0.8, 1.8)), float(rng.uniform(-2.0, 2.0))]
        inst.append(i)
    lights = []
    for k in range(int(rng.integers(0, 4))):
        kind = ["directional", "point", "spot"][int(rng.integers(0, 3))]
        li = {"type": kind, "name": "l%d" % k, "strength": f3(rng, 1.0, 15.0)}
        if kind != "point":
            li["direction"] = [float(rng.uniform(-1, 1)), float(rng.uniform(-1.0, -0.2)), float(rng.uniform(-1, 1))]
        if kind != "directional":
            li["position"] = [float(rng.uniform(-3, 3)), float(rng.uniform(1.5, 4)), float(rng.uniform(-3, 3))]
        if kind == "spot":
            li["inner_angle"], li["outer_angle"] = float(rng.uniform(5, 25)), float(rng.uniform(26, 60))
        lights.append(li)
    eye = np.array([rng.uniform(-3, 3), rng.uniform(0.5, 4.0), rng.uniform(4.0, 7.0)])
    fwd = np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.6, 0.4), 0.0]) - eye
    scene = {"cameras": {"type": "perspective", "name": "cam", "eye": [float(x) for x in eye], "forward": [float(x) for x in fwd], "up": [0.0, 1.0, 0.0],
                         "fov": float(rng.uniform(30, 70))},
             "textures": tex, "materials": mats, "mediums": mediums, "primitives": prims, "surfaces": surfaces, "instances": inst, "lights": lights}
    env = rng.random()
    if env < 0.35:
        scene["environment"] = {"type": "color", "color": f3(rng, 0.05, 0.6)}
    elif env < 0.55:
        scene["environment"] = {"type": "exr", "exr_file": "textures/env_small.exr", "scale": f3(rng, 0.3, 1.0)}
    if rng.random() < 0.3:
        scene["aggregate"] = "group"
    if rng.random() < 0.4:
        scene["light_sampler"] = "power_is"
    return scene


def stage_assets():
    work = tempfile.mkdtemp(prefix="spt_fuzz_")
    for sub in ("models", "textures"):
        shutil.copytree(os.path.join(SCENES, sub), os.path.join(work, sub))
    return work


def run_seed(seed, work):
    """-> (ok or None when the loader rejected the scene, one-line description, path of the scene JSON)"""
    rng = np.random.default_rng(1000 + seed)
    scene = make_scene(rng, work)
    path = os.path.join(work, "fuzz_%d.json" % seed)
    with open(path, "w") as fh:
        json.dump(scene, fh, indent=1)
    sampler = int(rng.integers(0, 3))
    dx, dy = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    spp = dx * dy if sampler == spt.SAMPLER_JITTERED else int(rng.integers(1, 9))
    radius = float(rng.choice([0.5, 0.5, 0.5, 0.3, 1.2, 1.6]))
    r = spt.PathTracer(max_depth=int(rng.integers(1, 9)), sampler=sampler, spp=spp, division_x=dx, division_y=dy, seed=int(rng.integers(0, 1 << 30)), filter_radius=radius)
    w, h = int(rng.integers(17, 120)), int(rng.integers(9, 90))
    # FUZZ_SIZE_MUL / FUZZ_SPP_MUL: the same scenes at larger images / sample counts (several wavefront passes, sample
    # chunks, full queues); the exhaustive oracle then takes seconds per seed
    size_mul, spp_mul = int(os.environ.get("FUZZ_SIZE_MUL", "1")), int(os.environ.get("FUZZ_SPP_MUL", "1"))
    w, h = w * size_mul, h * size_mul
    if sampler != spt.SAMPLER_JITTERED and spp_mul > 1:
        spp *= spp_mul
        r.spp = spp
    shard_count = int(rng.choice([1, 1, 2, 3]))
    strip_rows = int(rng.choice([1, 4, 16]))
    spp_pass = int(rng.integers(0, spp + 1))
    # FUZZ_SWITCHES=1: a random subset of the library's A/B switches per seed (read at scene creation / render time).
    # (SPT_WST_MASK and SPT_BEZ_DEFER belonged to two round-2 experiments that were removed in round 3 - the kind-sorted
    #  traversal and the deferred patch-pair pipeline; they stay in the list, ignored by the library, so that a seed still
    #  draws the same scene and the same other switches as when it was recorded)
    switches = {}
    if os.environ.get("FUZZ_SWITCHES"):
        for name, values in (("SPT_NO_FUSED", ["1"]), ("SPT_NO_LDS_TABLES", ["1"]), ("SPT_NO_LDS_GEO", ["1"]), ("SPT_NO_PIXEL_CULL", ["1"]),
                             ("SPT_NO_OVERLAP", ["1"]), ("SPT_NO_DYN_SHADOW", ["1"]), ("SPT_NO_DYN_EXTEND", ["1"]), ("SPT_PRIMARY_CHUNKS", ["1", "2", "7"]),
                             ("SPT_BOX_BAND_BYTES", ["20000", "300000"]), ("SPT_BVH_MAX_LEAF", ["1", "2", "8"]), ("SPT_DYN_BLOCKS", ["64", "512"])) + \
                            ((("SPT_NO_TAIL_LOOP", ["1"]), ("SPT_NO_STREAM", ["1"]), ("SPT_STREAM_MASK", ["0", "2", "7"]), ("SPT_STREAM_IFIF", ["0"]),
                              ("SPT_WST_MASK", ["1", "3"]), ("SPT_BEZ_LDS", ["1"]), ("SPT_BEZ_DEFER", ["1"])) if os.environ.get("FUZZ_V3") else ()) + \
                            ((("SPT_NO_CLASS_QUEUES", ["1"]), ("SPT_FLAT_BUDGET", ["0", "100000", "100000"]), ("SPT_NO_PACK_FIRST", ["1"]), ("SPT_NO_ROW_SPANS", ["1"]),
                              ("SPT_STREAM_REFILL", ["8", "64"])) if os.environ.get("FUZZ_V4") else ()) + \
                            ((("SPT_NO_EYE_BLOB", ["1"]),) if os.environ.get("FUZZ_V5") else ()):     # round 3 (V4), its last day (V5)
            os.environ.pop(name, None)
            if rng.random() < 0.25:
                switches[name] = values[int(rng.integers(0, len(values)))]
        if os.environ.get("SPT_REFERENCE_BVH") and (any(p["type"] in ("cubic_bezier", "catmull_clark") for p in scene["primitives"])):
            switches.pop("SPT_NO_LDS_GEO", None)     # patches + the caller's trees: only the exact LDS-resident nodes follow the reference's culling
        os.environ.update(switches)
    # FUZZ_V3: every third scene is loaded as the reference's `bezier_ni` build would see it (Newton's iteration on patches)
    newton = bool(os.environ.get("FUZZ_V3")) and seed % 3 == 0
    if newton:
        os.environ["SPT_BEZIER_NI"] = "1"
    try:
        sc = spt.load_scene(path)
    except spt.SptError as e:
        os.environ.pop("SPT_BEZIER_NI", None)
        return None, "scene rejected by the loader: %s" % str(e)[:120], path
    os.environ.pop("SPT_BEZIER_NI", None)
    ok, words, nan_px = True, 0, 0
    # (hundreds of patch instances: the exhaustive oracle would test every ray against every patch; the tree-walking one
    #  stands in - it can lose a ray that grazes the edge of an exact box, which the comparison would show)
    flags = _util.ORACLE_DEVICE if sc.desc.n_bezier_patches > 40 else _util.device_oracle_flags()
    v3 = bool(os.environ.get("FUZZ_V3"))
    for k in range(shard_count):
        ref, _ = _util.oracle_render(sc, r, w, h, flags=flags, shard_index=k, shard_count=shard_count, strip_rows=strip_rows)
        if v3 and radius == 0.5 and rng.random() < 0.5:
            # the asynchronous path: the frame is queued twice (the second copy-out waits for nothing but the first), then awaited
            r.render_shard(sc, spt.OutputConfig(w, h), shard_index=k, shard_count=shard_count, strip_rows=strip_rows, samples_per_pass=spp_pass, reuse_output=True)
            for _ in range(2):
                got = r.render_shard(sc, spt.OutputConfig(w, h), shard_index=k, shard_count=shard_count, strip_rows=strip_rows, samples_per_pass=spp_pass,
                                     reuse_output=True, wait=False)
            r.wait(sc)
            got = got.copy()
        else:
            got = r.render_shard(sc, spt.OutputConfig(w, h), shard_index=k, shard_count=shard_count, strip_rows=strip_rows, samples_per_pass=spp_pass)
        nan = np.isnan(ref)
        same_nan = np.array_equal(nan, np.isnan(got))
        diff = int((got.view(np.uint32) != ref.view(np.uint32))[~nan].sum())
        if (diff or not same_nan) and flags != _util.device_oracle_flags() and not os.environ.get("SPT_REFERENCE_BVH"):
            # the stand-in oracle walks the caller's EXACT boxes and can lose a grazing hit that the library's padded trees
            # keep (seed 625 of the round-2 campaign: one pixel): such a film is settled by the exhaustive oracle
            ref, _ = _util.oracle_render(sc, r, w, h, flags=_util.device_oracle_flags(), shard_index=k, shard_count=shard_count, strip_rows=strip_rows)
            nan = np.isnan(ref)
            same_nan = np.array_equal(nan, np.isnan(got))
            diff = int((got.view(np.uint32) != ref.view(np.uint32))[~nan].sum())
        ok = ok and same_nan and diff == 0
        words += diff
        nan_px += int(nan.any(axis=2).sum())
    # the intersection seams on adversarial rays: axis-aligned directions (zero components: infinite slab reciprocals),
    # rays that start exactly on a surface (the previous hit point), tiny and huge t ranges
    n = 3000
    rays = _util.random_rays(sc, n, seed=seed)
    ax = rng.integers(0, 3, n // 3)
    d = np.zeros((n // 3, 3), dtype=np.float32)
    d[np.arange(n // 3), ax] = rng.choice([-1.0, 1.0], n // 3)
    rays["d"][: n // 3] = d
    # (the rays always go against the exhaustive oracle: 6 000 rays x all patches is cheap, and axis-aligned rays are exactly
    #  the ones that graze the exact boxes the tree-walking oracle culls with)
    flags = _util.device_oracle_flags()
    first = _util.oracle_trace_closest(sc, rays, flags)
    hit = first["instance"] >= 0
    restart = rays.copy()
    restart["o"][hit] = (rays["o"][hit] + rays["d"][hit] * first["t"][hit][:, None]).astype(np.float32)
    nd = rng.normal(size=(n, 3)).astype(np.float32)
    restart["d"] = nd / np.linalg.norm(nd, axis=1, keepdims=True)
    restart["t_min"] = rng.choice([1e-4, 0.0, 1e-7, 1e-2], n).astype(np.float32)
    restart["t_max"] = rng.choice([3.4028234663852886e38, 1.0, 0.05, 30.0], n).astype(np.float32)
    ds = sc.device_scene(0)
    ray_bad = 0
    for batch in (rays, restart):
        ref_h = _util.oracle_trace_closest(sc, batch, flags)
        got_h = ds.trace_closest(batch)
        if os.environ.get("SPT_REFERENCE_BVH"):
            continue      # coincident surfaces make a few hits visit-order dependent in that mode (tests/test_gpu_parity.py)
        ray_bad += int(ref_h.tobytes() != got_h.tobytes())
        ray_bad += int(not np.array_equal(_util.oracle_trace_any(sc, batch, flags), ds.trace_any(batch)))
    ok = ok and ray_bad == 0
    kinds = sorted({m["type"] for m in scene["materials"]})
    info = ("%s  rays %s  %dx%d spp %d depth %d sampler %d radius %.1f shards %d/%d pass %d  inst %d lights %d env %s patches %d media %d  NaN px %d  words differ %d  %s" %
            ("ok  " if ok else "FAIL", "ok" if ray_bad == 0 else "BAD(%d)" % ray_bad, w, h, spp, r.max_depth, sampler, radius, shard_count, strip_rows, spp_pass, len(scene["instances"]), len(scene["lights"]),
             scene.get("environment", {}).get("type", "-"), sc.desc.n_bezier_patches, len(scene["mediums"]), nan_px, words, ",".join(k[:4] for k in kinds)))
    if newton and sc.desc.n_bezier_patches:
        info += "  newton"
    if switches:
        info += "  " + " ".join("%s=%s" % kv for kv in sorted(switches.items()))
    sc.close()
    return ok, info, path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:20")
    args = ap.parse_args()
    lo, hi = (int(x) for x in args.seeds.split(":"))
    out_dir = os.path.join(ROOT, "gpurun_out", "fuzz")
    work = stage_assets()
    bad = 0
    for seed in range(lo, hi):
        ok, info, path = run_seed(seed, work)
        print("seed %4d: %s" % (seed, info), flush=True)
        if ok is False:
            bad += 1
            os.makedirs(out_dir, exist_ok=True)
            shutil.copy(path, os.path.join(out_dir, os.path.basename(path)))
    shutil.rmtree(work, ignore_errors=True)
    print("%d of %d seeds failed" % (bad, hi - lo))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
