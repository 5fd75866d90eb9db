"""Image textures: the loader's PNG decode + mip pyramid + texture graph, and the oracle's
restatement of src/texture/*.rs, src/core/intersection.rs:28-84 and Surface::coord / emissive, against
independent numpy statements of the same rules and hand-computed values.  The reference ships no texture
assets or fixtures, so these known-answer tests are what pins this part ("parity unpinned" against the
Rust binary, like the rest of the path)."""
import ctypes as C
import json
import os
import struct
import zlib

import numpy as np
import pytest

import _util

spt = _util.load_pkg()


def png_bytes(width, height, depth, ctype, rows, palette=None, trns=None, filters=None):
    """Minimal PNG writer; `rows` are raw scanline bytes (already packed), `filters` per-row filter ids."""
    bpp = max(1, {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype] * depth // 8)
    raw = bytearray()
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        ft = filters[y] if filters else 0
        out = bytearray(len(row))
        for x in range(len(row)):
            a = row[x - bpp] if x >= bpp else 0
            b = prev[x]
            c = prev[x - bpp] if x >= bpp else 0
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = a
            elif ft == 2:
                pred = b
            elif ft == 3:
                pred = (a + b) >> 1
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out[x] = (row[x] - pred) & 255
        raw.append(ft)
        raw.extend(out)
        prev = bytes(row)

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xffffffff)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, ctype, 0, 0, 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    data = zlib.compress(bytes(raw))
    half = len(data) // 2
    out += chunk(b"IDAT", data[:half]) + chunk(b"IDAT", data[half:])    # split IDAT on purpose
    return out + chunk(b"IEND", b"")


def test_png_decode_all_colour_types_filters_and_depths(tmp_path):
    rng = np.random.default_rng(5)
    w, h = 7, 5
    # RGBA8 with every filter type
    px = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    (tmp_path / "rgba.png").write_bytes(png_bytes(w, h, 8, 6, [px[y].tobytes() for y in range(h)], filters=[0, 1, 2, 3, 4]))
    assert np.array_equal(spt.read_png(tmp_path / "rgba.png"), px)
    # RGB8 -> alpha 255
    (tmp_path / "rgb.png").write_bytes(png_bytes(w, h, 8, 2, [px[y, :, :3].tobytes() for y in range(h)], filters=[4, 3, 2, 1, 0]))
    got = spt.read_png(tmp_path / "rgb.png")
    assert np.array_equal(got[..., :3], px[..., :3]) and (got[..., 3] == 255).all()
    # gray8 and gray+alpha -> (l, l, l, a)
    (tmp_path / "g.png").write_bytes(png_bytes(w, h, 8, 0, [px[y, :, 0].tobytes() for y in range(h)]))
    got = spt.read_png(tmp_path / "g.png")
    assert all(np.array_equal(got[..., c], px[..., 0]) for c in range(3)) and (got[..., 3] == 255).all()
    (tmp_path / "ga.png").write_bytes(png_bytes(w, h, 8, 4, [px[y, :, :2].tobytes() for y in range(h)], filters=[1, 4, 1, 4, 2]))
    got = spt.read_png(tmp_path / "ga.png")
    assert np.array_equal(got[..., 0], px[..., 0]) and np.array_equal(got[..., 3], px[..., 1])
    # 16-bit RGB: (c + 128) / 257
    p16 = rng.integers(0, 65536, size=(h, w, 3), dtype=np.uint16)
    (tmp_path / "rgb16.png").write_bytes(png_bytes(w, h, 16, 2, [p16[y].astype(">u2").tobytes() for y in range(h)]))
    assert np.array_equal(spt.read_png(tmp_path / "rgb16.png")[..., :3], ((p16.astype(np.uint32) + 128) // 257).astype(np.uint8))
    # 2-bit palette with tRNS, width not a multiple of the packing
    idx = rng.integers(0, 4, size=(h, w))
    rows = []
    for y in range(h):
        bits = 0
        for x in range(8):
            bits = (bits << 2) | (int(idx[y, x]) if x < w else 0)
        rows.append(bits.to_bytes(2, "big"))
    pal = [10, 20, 30, 40, 50, 60, 70, 80, 90, 200, 210, 220]
    (tmp_path / "pal.png").write_bytes(png_bytes(w, h, 2, 3, rows, palette=pal, trns=[0, 128]))
    got = spt.read_png(tmp_path / "pal.png")
    assert np.array_equal(got[..., :3], np.array(pal, dtype=np.uint8).reshape(4, 3)[idx])
    assert np.array_equal(got[..., 3], np.array([0, 128, 255, 255], dtype=np.uint8)[idx])
    # 1-bit gray
    bitsrc = rng.integers(0, 2, size=(h, w))
    rows = [int("".join(str(int(b)) for b in list(bitsrc[y]) + [0]), 2).to_bytes(1, "big") for y in range(h)]
    (tmp_path / "g1.png").write_bytes(png_bytes(w, h, 1, 0, rows))
    assert np.array_equal(spt.read_png(tmp_path / "g1.png")[..., 0], (bitsrc * 255).astype(np.uint8))
    # errors: neither PNG nor JPEG (tests/test_jpeg.py covers JPEG), corrupt CRC, missing file
    (tmp_path / "x.gif").write_bytes(b"GIF89a" + bytes(32))
    with pytest.raises(spt.SptError) as e:
        spt.read_png(tmp_path / "x.gif")
    assert e.value.status == 103
    data = bytearray((tmp_path / "rgba.png").read_bytes())
    data[40] ^= 1
    (tmp_path / "bad.png").write_bytes(bytes(data))
    with pytest.raises(spt.SptError) as e:
        spt.read_png(tmp_path / "bad.png")
    assert e.value.status == 101
    with pytest.raises(spt.SptError) as e:
        spt.read_png(tmp_path / "nope.png")
    assert e.value.status == 100


def np_mip_chain(level0):
    """generate_mipmap (src/texture/image_tex.rs:66-100) in numpy."""
    levels = [level0]
    while levels[-1].shape[0] > 1 or levels[-1].shape[1] > 1:
        src = levels[-1].astype(np.float32)
        h, w = src.shape[:2]
        nh, nw = (h + 1) >> 1, (w + 1) >> 1
        y0 = 2 * np.arange(nh)
        y1 = np.minimum(y0 + 1, h - 1)
        x0 = 2 * np.arange(nw)
        x1 = np.minimum(x0 + 1, w - 1)
        s = ((src[y0][:, x0] + src[y1][:, x0]) + src[y0][:, x1]) + src[y1][:, x1]     # p0 + p1 + p2 + p3 in the reference's order
        levels.append((s * np.float32(0.25)).astype(np.uint8))                     # `as u8` truncates
    return levels


def scene_levels(sc, image):
    im = sc.array("images")[image]
    lv = sc.array("image_levels")[im["first_level"]:im["first_level"] + im["n_levels"]]
    tx = sc.array("texels")
    return [tx[l["first_texel"]:l["first_texel"] + l["width"] * l["height"]].view(np.uint8).reshape(l["height"], l["width"], 4) for l in lv]


@pytest.fixture(scope="module")
def textured():
    return spt.load_scene(os.path.join(_util.SCENES, "t_textured.json"))


def test_mip_pyramid_matches_generate_mipmap(textured):
    tex_dir = os.path.join(_util.SCENES, "textures")
    files = ["checker.png", "noise_rgba.png", "rough_ramp.png", "bumps_normal.png", "stripes_ga.png"]   # order of first use in the scene
    assert textured.desc.n_images == len(files)                      # one pyramid per file, shared by its textures
    for i, name in enumerate(files):
        want = np_mip_chain(spt.read_png(os.path.join(tex_dir, name)))
        got = scene_levels(textured, i)
        assert [g.shape for g in got] == [w.shape for w in want]
        assert got[-1].shape == (1, 1, 4)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)


def tex_named(scene_json, name):
    return next(t for t in scene_json["textures"] if t["name"] == name)


def node_of(sc, scene_json, name):
    """Index of the outermost node of the named texture = what a material referencing it points at."""
    # replay create_texture_from_params' node emission order: base, then srgb, then modifier
    n = 0
    for t in scene_json["textures"]:
        n += 1
        if t.get("is_srgb", False):
            n += 1
        if any(k not in ("type", "name", "value", "image_file", "t1", "t2", "is_srgb") and not k.startswith("#") for k in t):
            n += 1
        if t["name"] == name:
            return n - 1
    raise KeyError(name)


@pytest.fixture(scope="module")
def textured_json():
    with open(os.path.join(_util.SCENES, "t_textured.json")) as fh:
        return json.load(fh)


def test_texture_graph_nodes_and_recipes(textured, textured_json):
    tex = textured.array("textures")
    n = lambda name: node_of(textured, textured_json, name)
    # "checker": Modifier(Srgb(Image)) with tiling 3 and mirror_repeat
    m = tex[n("checker")]
    assert (m["type"], m["wrap"], m["mode"]) == (7, 1, -1) and np.allclose(m["tiling"], [3, 3, 1]) and np.allclose(m["offset"], 0)
    assert tex[m["a"]]["type"] == 6 and tex[tex[m["a"]]["a"]]["type"] == 1
    # position-mode modifier reads 3-component tiling / offset
    m = tex[n("noise_pos")]
    assert (m["mode"], m["wrap"]) == (2, 3) and np.allclose(m["tiling"], [0.7, 0.9, 1.0]) and np.allclose(m["offset"], [0.2, 0.1, 0.0])
    # binary ops point at the named (outermost) nodes
    mul = tex[n("noise_tinted")]
    assert mul["type"] == 4 and mul["a"] == n("noise") and mul["b"] == n("tint")
    sub = tex[n("one_minus_noise")]
    assert sub["type"] == 6 and tex[sub["a"]]["type"] == 3                       # is_srgb wraps the Sub node
    # constant materials stay folded, image-backed ones carry a recipe with the outermost nodes
    mats = textured.array("materials")
    rec = textured.array("material_recipes")
    assert (mats["recipe"] > 0).sum() == len(rec) == 8
    by_type = {int(r["type"]): r for r in rec}
    assert by_type[1]["tex"].tolist() == [n("eta_gold"), n("k_gold"), n("rough_map"), n("rough_map")]
    assert by_type[3]["tex"][0] == n("noise_tinted") and by_type[3]["tex"][2] == n("rough_plus") and abs(by_type[3]["ior"] - 1.5) < 1e-7
    assert by_type[4]["tex"].tolist() == [n("checker_clamp"), n("metal08"), n("rough_map"), n("rough_map")]
    surf = textured.array("surfaces")
    assert sorted(surf["normal_map"].tolist())[-2:] == [n("bumps") + 1] * 2 and surf["emissive_map"].max() == n("stripes") + 1
    # Surface::average_emissive feeds the shape light's power: emissive * image average (1x1 level)
    last = scene_levels(textured, 4)[-1][0, 0, :3].astype(np.float32) / 255.0
    lights = textured.array("lights")
    panel = lights[lights["type"] == 3][0]
    lum = float(np.dot(np.array([7.0, 6.5, 5.5]) * last, [0.299, 0.587, 0.114]))
    area = 4.0 * 1.4 * 0.9                                                     # 2x2 plane scaled by (1.4, 0.9)
    assert abs(panel["power"] - area * lum) < 1e-3 * area * lum
    # errors the reference raises too
    for mutate, msg in ((lambda s: tex_named(s, "checker").update(mode="spherical"), "Unknown texture input mode"),
                        (lambda s: tex_named(s, "checker").update(wrap="border"), "Unknown texture input wrap mode"),
                        (lambda s: tex_named(s, "checker").update(image_file="textures/nope.png"), "can't read image")):
        bad = json.loads(json.dumps(textured_json))
        mutate(bad)
        path = os.path.join(_util.SCENES, "_tmp_bad_textured.json")
        with open(path, "w") as fh:
            json.dump(bad, fh)
        try:
            with pytest.raises(spt.SptError) as e:
                spt.load_scene(path)
            assert msg in str(e.value), str(e.value)
        finally:
            os.remove(path)


def np_bilinear(level, u, v):
    """sample_blinear (image_tex.rs:102-125) in f32 numpy."""
    f = np.float32
    h, w = level.shape[:2]
    x = f(u) * f(w)
    x1 = int(np.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)
    x0 = x1 - 1
    xt = f(f(x - f(x0)) - f(0.5))
    y = f(v) * f(h)
    y1 = int(np.floor(abs(y) + 0.5)) * (1 if y >= 0 else -1)
    y0 = y1 - 1
    yt = f(f(y - f(y0)) - f(0.5))
    cx = lambda i: min(max(i, 0), w - 1)
    cy = lambda i: min(max(i, 0), h - 1)
    px = lambda xx, yy: level[cy(yy), cx(xx)].astype(np.float32) / f(255.0)
    c0 = px(x0, y0) * f(f(1.0) - yt) + px(x0, y1) * yt
    c1 = px(x1, y0) * f(f(1.0) - yt) + px(x1, y1) * yt
    return c0 * f(f(1.0) - xt) + c1 * xt


def test_image_sampling_bilinear_trilinear_and_alpha(textured, textured_json):
    n = lambda name: node_of(textured, textured_json, name)
    levels = scene_levels(textured, 1)                       # noise_rgba.png 33x17, all four channels random
    raw = n("noise") - 1                                     # the bare Image node under the tiling modifier
    assert textured.array("textures")[raw]["type"] == 1
    rng = np.random.default_rng(3)
    uv = rng.random((200, 2)).astype(np.float32)
    got = _util.oracle_tex_eval(textured, raw, uv)
    want = np.stack([np_bilinear(levels[0], u, v) for u, v in uv])
    assert np.array_equal(got.view(np.uint32), want.astype(np.float32).view(np.uint32))      # zero differentials: level 0, bit-exact
    # texel centres reproduce the texels exactly
    h, w = levels[0].shape[:2]
    centres = np.array([[(x + 0.5) / w, (y + 0.5) / h] for x, y in ((0, 0), (5, 3), (32, 16), (17, 9))], dtype=np.float32)
    got = _util.oracle_tex_eval(textured, raw, centres)
    for (x, y), g in zip(((0, 0), (5, 3), (32, 16), (17, 9)), got):
        assert np.allclose(g, levels[0][y, x].astype(np.float32) / 255.0, atol=2e-6)
    # trilinear: level = clamp(log2(max(|duvdx * size|, |duvdy * size|) + 0.001), 0, n - 1)
    for footprint, lo, hi in ((1.0, 0, 0), (2.0, 1, 1), (3.0, 1, 2), (5.0, 2, 3), (1000.0, 6, 6)):
        duvdx = (footprint / w, 0.0)
        got = _util.oracle_tex_eval(textured, raw, uv[:20], duvdx=duvdx, duvdy=(0.0, 0.5 / h))
        level = float(np.clip(np.log2(footprint + 0.001), 0, len(levels) - 1))
        l0 = int(np.floor(level))
        assert lo <= l0 <= hi
        for (u, v), g in zip(uv[:20], got):
            a = np_bilinear(levels[l0], u, v)
            b = np_bilinear(levels[min(l0 + 1, len(levels) - 1)], u, v)
            lt = np.float32(level - l0)
            assert np.allclose(g, a * (1 - lt) + b * lt, atol=3e-6)
    # the 1x1 level is the constant the pyramid converged to
    got = _util.oracle_tex_eval(textured, raw, uv[:5], duvdx=(10.0, 0.0))
    assert np.allclose(got, levels[-1][0, 0].astype(np.float32) / 255.0, atol=1e-6)


def test_wrap_modes_tiling_offset_and_input_modes(textured, textured_json):
    n = lambda name: node_of(textured, textured_json, name)
    tex = textured.array("textures")
    lv_checker = scene_levels(textured, 0)[0]
    lv_noise = scene_levels(textured, 1)[0]
    # Clamp with tiling 1.6 and offset -0.3 (checker_clamp): u' = clamp(u * 1.6 - 0.3, 0, 1)
    uv = np.array([[0.0, 0.0], [0.1, 0.9], [0.5, 0.5], [0.95, 0.2], [-3.0, 7.0]], dtype=np.float32)
    got = _util.oracle_tex_eval(textured, n("checker_clamp"), uv)
    for (u, v), g in zip(uv, got):
        uu = np.clip(np.float32(u) * np.float32(1.6) + np.float32(-0.3), 0, 1)
        vv = np.clip(np.float32(v) * np.float32(1.6) + np.float32(-0.3), 0, 1)
        assert np.allclose(g, np_bilinear(lv_checker, uu, vv), atol=1e-6)
    # MirrorRepeat with tiling 3 under an sRGB decode (checker): even periods forward, odd periods mirrored
    def mirror(x):
        fr = x - np.trunc(x)
        xn = fr if x >= 0 else 1 + fr
        return xn if int(x) % 2 == 0 else 1 - xn
    def srgb(c):
        c = np.asarray(c, dtype=np.float64)
        return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    uv = np.array([[0.1, 0.2], [0.4, 0.7], [0.9, 0.55], [-0.2, 1.3]], dtype=np.float32)
    got = _util.oracle_tex_eval(textured, n("checker"), uv)
    for (u, v), g in zip(uv, got):
        want = np_bilinear(lv_checker, np.float32(mirror(float(np.float32(u) * 3))), np.float32(mirror(float(np.float32(v) * 3))))
        assert np.allclose(g[:3], srgb(want[:3]), atol=2e-6) and abs(g[3] - want[3]) < 1e-6      # alpha is not decoded
    # Repeat (default) with tiling (2, 1): negative coordinates wrap to 1 + fract
    uv = np.array([[0.3, 0.4], [0.8, 0.4], [-0.15, -0.25]], dtype=np.float32)
    got = _util.oracle_tex_eval(textured, n("noise"), uv)
    rep = lambda x: (x - np.trunc(x)) if x >= 0 else 1 + (x - np.trunc(x))
    for (u, v), g in zip(uv, got):
        assert np.allclose(g, np_bilinear(lv_noise, np.float32(rep(float(np.float32(u) * 2))), np.float32(rep(float(v)))), atol=1e-6)
    # mode "position" + mirror_clamp: reads position.xy * tiling + offset, texcoords are ignored
    pos = np.array([[0.5, 0.25, 9.0], [2.0, -1.0, 0.0], [-0.1, 0.6, 3.0]], dtype=np.float32)
    got = _util.oracle_tex_eval(textured, n("noise_pos"), np.full((3, 2), 0.77, dtype=np.float32), position=pos)
    for p, g in zip(pos, got):
        uu = abs(np.clip(np.float32(p[0]) * np.float32(0.7) + np.float32(0.2), 0, 1))
        vv = abs(np.clip(np.float32(p[1]) * np.float32(0.9) + np.float32(0.1), 0, 1))
        assert np.allclose(g, np_bilinear(lv_noise, uu, vv), atol=1e-6)
    assert tex[n("noise_pos")]["mode"] == 2


def test_binary_ops_srgb_and_scalar_alpha(textured, textured_json):
    n = lambda name: node_of(textured, textured_json, name)
    uv = np.array([[0.21, 0.43], [0.77, 0.05]], dtype=np.float32)
    noise = _util.oracle_tex_eval(textured, n("noise"), uv)
    tint = _util.oracle_tex_eval(textured, n("tint"), uv)
    assert np.allclose(tint, [0.9, 0.7, 0.5, 1.0])                                    # ScalarTex: alpha 1
    mul = _util.oracle_tex_eval(textured, n("noise_tinted"), uv)
    assert np.array_equal(mul.view(np.uint32), (noise * tint).view(np.uint32))
    rough = _util.oracle_tex_eval(textured, n("rough_map"), uv)
    add = _util.oracle_tex_eval(textured, n("rough_plus"), uv)
    assert np.array_equal(add[:, 0].view(np.uint32), (rough[:, 0] + np.float32(0.3)).view(np.uint32))
    sub = _util.oracle_tex_eval(textured, n("one_minus_noise"), uv)                    # Srgb(white - noise_tinted)
    lin = (np.float32(0.8) - mul[:, :3]).astype(np.float64)
    want = np.where(lin <= 0.04045, lin / 12.92, ((np.maximum(lin, 0) + 0.055) / 1.055) ** 2.4)
    assert np.allclose(sub[:, :3], want, atol=3e-6)
    half = _util.oracle_tex_eval(textured, n("checker_half"), uv)                      # Div under a wrap-only modifier
    clamp = _util.oracle_tex_eval(textured, n("checker_clamp"), uv)
    assert np.array_equal(half[:, :3].view(np.uint32), (clamp[:, :3] / np.float32(0.5)).view(np.uint32))
    # deterministic pow agrees with libm's powf to a few ulp
    libm = _util.oracle_tex_eval(textured, n("checker"), uv, flags=_util.ORACLE_LIBM)
    det = _util.oracle_tex_eval(textured, n("checker"), uv)
    assert np.allclose(libm, det, rtol=2e-6, atol=1e-7)


def test_calc_differential_on_analytic_planes():
    lib = _util.oracle_lib()

    def diff(o, d, xo, xd, yo, yd, t, normal, tangent, bitangent):
        ray = np.array(list(o) + list(d) + list(xo) + list(xd) + list(yo) + list(yd), dtype=np.float32)
        hit = np.array([t] + list(normal) + list(tangent) + list(bitangent), dtype=np.float32)
        dx, dy = np.zeros(2, dtype=np.float32), np.zeros(2, dtype=np.float32)
        lib.oracle_calc_differential(ray.ctypes.data, hit.ctypes.data, dx.ctypes.data, dy.ctypes.data)
        return dx, dy
    # plane z = 0 parametrised by (u, v) = (x, y): offsets of the auxiliary rays ARE the uv differentials
    o = (0.2, 0.3, 2.0)
    dx, dy = diff(o, (0, 0, -1), (0.25, 0.3, 2.0), (0, 0, -1), (0.2, 0.37, 2.0), (0, 0, -1), 2.0, (0, 0, 1), (1, 0, 0), (0, 1, 0))
    assert np.allclose(dx, [0.05, 0.0], atol=1e-6) and np.allclose(dy, [0.0, 0.07], atol=1e-6)
    # dominant-axis selection: the same plane rotated to x = 0 (tangent +y, bitangent +z) and to y = 0 (tangent +z, bitangent +x)
    dx, dy = diff((2.0, 0.2, 0.3), (-1, 0, 0), (2.0, 0.25, 0.3), (-1, 0, 0), (2.0, 0.2, 0.37), (-1, 0, 0), 2.0, (1, 0, 0), (0, 1, 0), (0, 0, 1))
    assert np.allclose(dx, [0.05, 0.0], atol=1e-6) and np.allclose(dy, [0.0, 0.07], atol=1e-6)
    dx, dy = diff((0.3, 2.0, 0.2), (0, -1, 0), (0.3, 2.0, 0.25), (0, -1, 0), (0.37, 2.0, 0.2), (0, -1, 0), 2.0, (0, 1, 0), (0, 0, 1), (1, 0, 0))
    assert np.allclose(dx, [0.05, 0.0], atol=1e-6) and np.allclose(dy, [0.0, 0.07], atol=1e-6)
    # stretched parametrisation (tangent length 2 = half the uv rate) and perspective rays from one eye
    eye = (0.0, 0.0, 4.0)
    d0 = np.array([0.1, -0.05, -1.0]); d0 /= np.linalg.norm(d0)
    dxr = np.array([0.11, -0.05, -1.0]); dxr /= np.linalg.norm(dxr)
    dyr = np.array([0.1, -0.04, -1.0]); dyr /= np.linalg.norm(dyr)
    t = 4.0 / -d0[2]
    dx, dy = diff(eye, d0, eye, dxr, eye, dyr, t, (0, 0, 1), (2, 0, 0), (0, 1, 0))
    px = np.array(eye) + dxr * (4.0 / -dxr[2]) - (np.array(eye) + d0 * t)
    py = np.array(eye) + dyr * (4.0 / -dyr[2]) - (np.array(eye) + d0 * t)
    assert np.allclose(dx, [px[0] / 2, px[1]], atol=1e-5) and np.allclose(dy, [py[0] / 2, py[1]], atol=1e-5)
    # singular frame: no solution, differentials stay zero (intersection.rs:76-82)
    dx, dy = diff(o, (0, 0, -1), (0.25, 0.3, 2.0), (0, 0, -1), (0.2, 0.37, 2.0), (0, 0, -1), 2.0, (0, 0, 1), (1, 0, 0), (2, 0, 0))
    assert np.array_equal(dx, [0, 0]) and np.array_equal(dy, [0, 0])


def test_textured_render_golden_properties(textured):
    """End-to-end through the oracle: the textures must actually drive the image (a flat-albedo render differs), the
    emissive map modulates the panel seen from below, and mip selection blurs the distant floor."""
    r = spt.PathTracer(max_depth=6, sampler=spt.SAMPLER_RECURRENCE, spp=16, seed=11)
    film, _ = _util.oracle_render(textured, r, 96, 72, flags=_util.ORACLE_DEVICE)
    ok = np.isfinite(film).all(axis=2)
    assert ok.mean() > 0.999
    floor_rows = film[60:70, 20:80][ok[60:70, 20:80]]
    assert floor_rows.std(axis=0).max() > 0.02                 # checker contrast on the near floor
    assert 0.05 < float(np.nanmean(film)) < 1.0
