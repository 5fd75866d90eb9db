"""glTF 2.0 import (role of src/loader/gltf.rs + the `gltf` crate's import()): the committed
asset scenes_amd/t_gltf.gltf (written by scenes_amd/make_scenes.py) and small in-test documents."""
import base64
import json
import os
import struct

import numpy as np
import pytest

import _util

spt = _util.load_pkg()
ASSET = os.path.join(_util.SCENES, "t_gltf.gltf")


@pytest.fixture(scope="module")
def scene():
    return spt.load_scene(ASSET)


def test_meshes_instances_and_node_transforms(scene):
    d = scene.desc
    assert (d.n_meshes, d.n_instances, d.n_tris) == (4, 4, 2 + 12 + 12 + 2)
    inst = scene.array("instances")
    by_tris = {int(scene.array("meshes")[i["prim_id"]]["tri_count"]): i for i in inst}
    # the "floor" node: T(0,-1,0) * S(4,1,4) on the unit quad
    floor = [i for i in inst if scene.array("meshes")[i["prim_id"]]["tri_count"] == 2 and abs(i["fwd"][10] + 1.0) < 1e-6][0]
    fwd = floor["fwd"].reshape(4, 3)
    assert np.allclose(fwd[:3].T, np.diag([4.0, 1.0, 4.0])) and np.allclose(fwd[3], [0.0, -1.0, 0.0])
    assert np.allclose(floor["bmin"], [-4, -1, -4], atol=1e-5) and np.allclose(floor["bmax"], [4, -1, 4], atol=1e-5)
    # child nodes inherit the parent's TRS: group = T(-0.8,-0.5,0.2) * R_y(45 deg); second child adds its own matrix
    c, s = np.cos(np.pi / 4), np.sin(np.pi / 4)
    Ry = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    boxes = [i for i in inst if scene.array("meshes")[i["prim_id"]]["tri_count"] == 12]
    assert len(boxes) == 2
    plain = [b for b in boxes if scene.array("materials")[scene.array("surfaces")[b["surface"]]["material"]]["recipe"] == 0][0]
    textured = [b for b in boxes if b is not plain][0]
    assert np.allclose(textured["fwd"].reshape(4, 3)[:3].T, Ry, atol=1e-6) and np.allclose(textured["fwd"].reshape(4, 3)[3], [-0.8, -0.5, 0.2], atol=1e-6)
    M = Ry @ np.diag([0.6, 0.6, 0.6])
    t = Ry @ np.array([1.6, -0.2, 0.5]) + np.array([-0.8, -0.5, 0.2])
    assert np.allclose(plain["fwd"].reshape(4, 3)[:3].T, M, atol=1e-6) and np.allclose(plain["fwd"].reshape(4, 3)[3], t, atol=1e-6)
    # mesh without NORMAL: TriMesh::calc_normals gives the face normals of the box (each face has its own 4 vertices)
    m = scene.array("meshes")[plain["prim_id"]]
    attr = scene.array("tri_attr")[m["tri_first"]:m["tri_first"] + m["tri_count"]]
    pos = scene.array("tri_pos")[m["tri_first"]:m["tri_first"] + m["tri_count"]]
    for a, p in zip(attr, pos):
        n = np.cross(p["p1"] - p["p0"], p["p2"] - p["p0"])
        n /= np.linalg.norm(n)
        assert np.allclose(a["n"], n, atol=1e-6)


def test_materials_follow_gltf_rs(scene):
    mats, rec, tex = scene.array("materials"), scene.array("material_recipes"), scene.array("textures")
    assert len(mats) == 4 and len(rec) == 2
    # metallic-roughness: base = Mul(Scalar(factor), Srgb(image_0)); metallic / roughness = Mul(Scalar(f), image_1), channels B / G
    mr = rec[mats[0]["recipe"] - 1]
    assert (mr["type"], mr["rough_chan"], mr["metal_chan"]) == (4, 1, 2)
    base, metal, rough = tex[mr["tex"][0]], tex[mr["tex"][1]], tex[mr["tex"][2]]
    assert base["type"] == 4 and np.allclose(tex[base["a"]]["value"], [0.9, 0.85, 0.8]) and tex[base["b"]]["type"] == 6
    assert tex[tex[base["b"]]["a"]]["type"] == 1 and tex[tex[base["b"]]["a"]]["image"] == 0
    assert metal["type"] == 4 and np.allclose(tex[metal["a"]]["value"], 0.6) and tex[metal["b"]]["image"] == 1
    assert rough["type"] == 4 and np.allclose(tex[rough["a"]]["value"], 0.9) and mr["tex"][2] == mr["tex"][3]
    # specular-glossiness: roughness = Sub(scalar_one, Mul(Scalar(gloss), image_1)) read from ALPHA
    sg = rec[mats[1]["recipe"] - 1]
    assert (sg["type"], sg["rough_chan"]) == (5, 3)
    r = tex[sg["tex"][2]]
    assert r["type"] == 3 and np.allclose(tex[r["a"]]["value"], 1.0) and tex[r["b"]]["type"] == 4
    assert np.allclose(tex[tex[r["b"]]["a"]]["value"], 0.8) and tex[tex[r["b"]]["b"]]["image"] == 1
    # constant materials are folded: plain = PbrMetallic(base (0.2,0.5,0.8), metallic 0, roughness 0.5 -> alpha 0.25)
    assert mats[2]["recipe"] == 0 and mats[2]["bxdf"] == 6 and np.allclose(mats[2]["c0"], [0.2, 0.5, 0.8]) and np.allclose(mats[2]["c1"], 0.04)
    assert abs(mats[2]["ax"] - 0.25) < 1e-7 and mats[2]["fresnel"] == 1 and mats[2]["substrate"] == 0
    surf = scene.array("surfaces")
    assert surf["flags"].tolist() == [0, 1, 0, 1]                      # doubleSided
    assert surf[0]["normal_map"] > 0 and surf[3]["emissive_map"] > 0 and np.allclose(surf[3]["emissive"], [4.0, 3.5, 3.0])


def test_camera_and_punctual_lights(scene):
    cam = scene.get_camera("cam")
    a = np.radians(15.0)                                               # rotation of -15 deg about x
    assert np.allclose(list(cam.eye), [0.0, 1.2, 5.5]) and np.allclose(list(cam.forward), [0.0, -np.sin(a), -np.cos(a)], atol=1e-6)
    assert abs(cam.half_cot_half_fov - 0.5 / np.tan(0.35)) < 1e-6      # yfov is in radians
    lights = scene.array("lights")
    assert sorted(lights["type"].tolist()) == [0, 1, 2, 3]             # directional, point, spot + the emissive panel
    sun = lights[lights["type"] == 0][0]
    assert abs(np.linalg.norm(sun["dir"]) - 1.0) < 1e-6 and np.allclose(sun["strength"], np.array([1.0, 0.95, 0.9]) * 1.5)
    point = lights[lights["type"] == 1][0]
    assert np.allclose(point["pos"], [1.5, 1.0, 1.0]) and np.allclose(point["strength"], np.array([0.3, 0.5, 1.0]) * 6.0)
    spot = lights[lights["type"] == 2][0]
    assert np.allclose(spot["dir"], [0.0, -1.0, 0.0], atol=1e-6) and np.allclose([spot["cos_inner"], spot["cos_outer"]], np.cos([0.3, 0.6]))


def _tiny_gltf(tmp_path, mutate=None, glb=False):
    pos = struct.pack("<9f", 0, 0, 0, 1, 0, 0, 0, 1, 0)
    idx = struct.pack("<3H", 0, 1, 2) + b"\0\0"
    blob = pos + idx
    doc = {"asset": {"version": "2.0"},
           "buffers": [{"byteLength": len(blob)}] if glb else [{"uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode(), "byteLength": len(blob)}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 6}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}, {"bufferView": 1, "componentType": 5123, "count": 3, "type": "SCALAR"}],
           "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.5, 0.5, 0.5, 1.0]}}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1, "material": 0}]}],
           "cameras": [{"type": "perspective", "perspective": {"yfov": 1.0, "znear": 0.1}}],
           "nodes": [{"mesh": 0}, {"camera": 0, "translation": [0.0, 0.0, 3.0]}],
           "scenes": [{"nodes": [0, 1]}]}
    if mutate:
        mutate(doc)
    if glb:
        js = json.dumps(doc).encode()
        js += b" " * (-len(js) % 4)
        body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(blob), 0x004E4942) + blob
        path = tmp_path / "t.glb"
        path.write_bytes(b"glTF" + struct.pack("<II", 2, 12 + len(body)) + body)
    else:
        path = tmp_path / "t.gltf"
        path.write_text(json.dumps(doc))
    return str(path)


def test_embedded_buffers_glb_container_and_defaults(tmp_path):
    for glb in (False, True):
        sc = spt.load_scene(_tiny_gltf(tmp_path, glb=glb))
        assert (sc.desc.n_instances, sc.desc.n_tris, sc.desc.aggregate, sc.desc.light_sampler) == (1, 1, 1, 0)
        m = sc.array("materials")[0]       # defaults: metallicFactor 1, roughnessFactor 1 -> all specular colour, no diffuse
        assert np.allclose(m["c1"], 0.5) and np.allclose(m["c0"], 0.0) and abs(m["ax"] - 1.0) < 1e-7
        assert sc.get_camera("camera_1").half_cot_half_fov == pytest.approx(0.5 / np.tan(0.5))
        assert np.allclose(sc.array("tri_attr")[0]["n"], [0, 0, 1])   # calc_normals


def test_errors_and_reference_quirks(tmp_path):
    def no_material(d):
        del d["meshes"][0]["primitives"][0]["material"]
    def no_indices(d):
        del d["meshes"][0]["primitives"][0]["indices"]
    def u8_indices(d):
        d["accessors"][1]["componentType"] = 5121
    def strided(d):
        d["bufferViews"][0]["byteStride"] = 16
    def tex_index_is_image_index(d):   # texture 1 -> source 0, but the reference looks up image_1
        png = open(os.path.join(_util.SCENES, "textures", "checker.png"), "rb").read()
        d["images"] = [{"uri": "data:image/png;base64," + base64.b64encode(png).decode()}]
        d["textures"] = [{"source": 0}, {"source": 0}]
        d["materials"][0]["pbrMetallicRoughness"]["baseColorTexture"] = {"index": 1}
    for mutate, status, msg in ((no_material, 102, "no material"), (no_indices, 102, "doesn't have indices"), (u8_indices, 103, "u16 / u32"),
                                (strided, 103, "byteStride"), (tex_index_is_image_index, 102, "image_1")):
        with pytest.raises(spt.SptError) as e:
            spt.load_scene(_tiny_gltf(tmp_path, mutate))
        assert e.value.status == status and msg in str(e.value), str(e.value)
    with pytest.raises(spt.SptError) as e:          # no camera node: to_scene fails like for JSON scenes
        spt.load_scene(_tiny_gltf(tmp_path, lambda d: d["scenes"][0].update(nodes=[0])))
    assert "At least one camera" in str(e.value)


def test_gltf_key_of_a_json_scene_merges_without_overriding(tmp_path):
    base = {"cameras": {"type": "perspective", "name": "cam", "eye": [0.0, 0.0, 9.0], "forward": [0.0, 0.0, -1.0], "up": [0.0, 1.0, 0.0], "fov": 30.0},
            "textures": [{"type": "scalar", "name": "w", "value": [1.0, 1.0, 1.0]}],
            "materials": [{"type": "lambert", "name": "m", "albedo": "w"}], "mediums": [], "surfaces": [],
            "primitives": [{"type": "sphere", "name": "s", "radius": 1.0}],
            "instances": [{"name": "ball", "primitive": "s", "material": "m"}],
            "lights": [{"type": "directional", "name": "sun", "direction": [0.0, -1.0, 0.0], "strength": [9.0, 9.0, 9.0]}],
            "gltf": os.path.relpath(ASSET, str(tmp_path))}
    p = tmp_path / "scene.json"
    p.write_text(json.dumps(base))
    sc = spt.load_scene(str(p))
    assert sc.desc.n_instances == 5 and sc.desc.n_spheres == 1
    lights = sc.array("lights")
    suns = lights[lights["type"] == 0]
    assert len(suns) == 1 and np.allclose(suns[0]["strength"], 9.0)        # JSON "sun" wins over the glTF node "sun"
    assert list(sc.get_camera("cam").eye) == [0.0, 0.0, 9.0]               # JSON "cam" wins over the glTF node "cam"


def test_oracle_renders_the_asset(scene):
    r = spt.PathTracer(max_depth=6, sampler=spt.SAMPLER_RECURRENCE, spp=16, seed=2)
    film, _ = _util.oracle_render(scene, r, 96, 72, flags=_util.ORACLE_DEVICE)
    assert np.isfinite(film).all() and 0.005 < float(film.mean()) < 1.0
    assert film[50:70, 20:76].std() > 0.005                                 # textured floor is visible
