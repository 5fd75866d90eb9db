"""JPEG textures in, JPEG images out (host side; reference: `image::open` in src/core/loader.rs:366-371 and
`image.save` in src/renderer/pt.rs:292-294).

The decoder restates the IJG arithmetic and is pinned to libjpeg-turbo bit-for-bit by the committed fixture
tests/golden/jpeg_cases.npz (written by tests/make_jpeg_golden.py through Pillow).  Against the reference's Rust
`jpeg-decoder` crate the texels are "parity unpinned" (same algorithm, rounding not guaranteed identical)."""
import json
import os

import numpy as np
import pytest

import _util

spt = _util.load_pkg()
G = np.load(os.path.join(_util.GOLDEN, "jpeg_cases.npz"))
NAMES = sorted(k[:-4] for k in G.files if k.endswith("_jpg"))


@pytest.mark.parametrize("name", NAMES)
def test_decoder_equals_libjpeg(name, tmp_path):
    p = tmp_path / (name + ".jpg")
    p.write_bytes(G[name + "_jpg"].tobytes())
    got = spt.read_png(str(p))
    assert got.shape[:2] == G[name + "_rgb"].shape[:2] and (got[..., 3] == 255).all()
    assert np.array_equal(got[..., :3], G[name + "_rgb"])


def test_decoder_against_pillow_when_available(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    y, x = np.mgrid[0:75, 0:83]
    img = np.clip(np.stack([x * 3, y * 3, (x * y) % 256], axis=-1) + rng.normal(0, 20, (75, 83, 3)), 0, 255).astype(np.uint8)
    for k, kw in enumerate([dict(quality=q, subsampling=s, progressive=pr) for q in (25, 88) for s in (0, 1, 2) for pr in (False, True)]):
        p = str(tmp_path / ("c%d.jpg" % k))
        Image.fromarray(img).save(p, **kw)
        assert np.array_equal(spt.read_png(p)[..., :3], np.asarray(Image.open(p).convert("RGB"))), kw


def test_bad_files_are_reported(tmp_path):
    data = G["base_420_odd_jpg"].tobytes()
    for cut, what in ((len(data) // 2, None), (20, None)):
        p = tmp_path / "cut.jpg"
        p.write_bytes(data[:cut])
        try:
            spt.read_png(str(p))          # a truncated scan decodes to padding (zeros are fed), as libjpeg does with a warning
        except spt.SptError as e:
            assert "jpeg" in str(e)
    p = tmp_path / "arith.jpg"
    p.write_bytes(data.replace(b"\xff\xc0", b"\xff\xc9", 1))
    with pytest.raises(spt.SptError) as e:
        spt.read_png(str(p))
    assert "arithmetic" in str(e.value)
    p = tmp_path / "x.gif"
    p.write_bytes(b"GIF89a" + bytes(64))
    with pytest.raises(spt.SptError) as e:
        spt.read_png(str(p))
    assert "PNG and JPEG" in str(e.value)


def test_encoder_roundtrip_and_extension_dispatch(tmp_path):
    y, x = np.mgrid[0:45, 0:70]
    rgb = np.stack([x * 3 % 256, y * 5 % 256, (x + y) * 2 % 256], axis=-1).astype(np.uint8)
    for q, floor in ((75, 30.0), (95, 36.0)):
        p = str(tmp_path / "o.jpg")
        spt.write_jpeg(p, rgb, q)
        back = spt.read_png(p)[..., :3].astype(np.float64)
        psnr = 10 * np.log10(255.0**2 / ((back - rgb) ** 2).mean())
        assert psnr > floor, psnr
    film = rgb.astype(np.float32) / np.float32(255.0)
    spt.write_image(str(tmp_path / "a.png"), film)
    spt.write_image(str(tmp_path / "a.JPEG"), film)
    assert np.array_equal(spt.read_png(str(tmp_path / "a.png"))[..., :3], spt.film_to_rgb8(film))
    assert spt.read_png(str(tmp_path / "a.JPEG")).shape == (45, 70, 4)
    with pytest.raises(spt.SptError) as e:
        spt.write_image(str(tmp_path / "a.bmp"), film)
    assert "extension" in str(e.value)


def test_jpeg_texture_in_a_scene_equals_the_same_pixels_as_png(tmp_path):
    """an `image_file` texture decodes a .jpg by content; the scene built from it equals the one built from a PNG of
    the decoded pixels (mips, sRGB flag and all)"""
    name = "base_420_odd"
    (tmp_path / "t.jpg").write_bytes(G[name + "_jpg"].tobytes())
    rgb = G[name + "_rgb"]
    spt.write_image(str(tmp_path / "t.png"), rgb.astype(np.float32) / np.float32(255.0) + np.float32(0.001))
    assert np.array_equal(spt.read_png(str(tmp_path / "t.png"))[..., :3], rgb)
    scenes = []
    for ext in ("jpg", "png"):
        sc = {
            "cameras": [{"type": "perspective", "name": "c", "eye": [0.0, 0.0, 3.0], "forward": [0.0, 0.0, -1.0], "up": [0.0, 1.0, 0.0], "fov": 40.0}],
            "textures": [{"type": "image", "name": "img", "image_file": "t." + ext}],
            "materials": [{"type": "lambert", "name": "m", "albedo": "img"}],
            "mediums": [], "primitives": [{"type": "sphere", "name": "s", "radius": 1.0}],
            "surfaces": [{"name": "sf", "material": "m"}],
            "instances": [{"name": "i", "primitive": "s", "surface": "sf"}],
            "lights": [{"type": "directional", "name": "sun", "direction": [0.0, -1.0, -1.0], "strength": [2.0, 2.0, 2.0]}],
        }
        p = tmp_path / ("s_%s.json" % ext)
        p.write_text(json.dumps(sc))
        scenes.append(spt.load_scene(str(p)))
    a, b = scenes
    assert a.desc.n_texels == b.desc.n_texels > 0
    assert np.array_equal(a.array("texels"), b.array("texels"))
