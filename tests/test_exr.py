"""OpenEXR scanline reader of the host library (`environment {type: "exr"}`, reference get_exr_image in
src/core/loader.rs:374-390 through the `exr` crate): every lossless / reader-deterministic compression the library
accepts, written here by an independent Python encoder of the file format (ImfRle / ImfZip / ImfPxr24Compressor
layouts).  No EXR tool exists in the image and the reference ships no EXR file, so these are format-spec tests:
"parity unpinned" against files written by OpenEXR itself."""
import struct
import zlib

import numpy as np
import pytest

import _util

spt = _util.load_pkg()
NONE, RLE, ZIPS, ZIP, PIZ, PXR24 = 0, 1, 2, 3, 4, 5


def _attr(name, ty, data):
    return name.encode() + b"\0" + ty.encode() + b"\0" + struct.pack("<i", len(data)) + data


def _predict_and_split(raw):
    """the byte re-ordering + delta predictor ZIP and RLE share (ImfZip.cpp / ImfRleCompressor.cpp)"""
    raw = np.frombuffer(raw, dtype=np.uint8)
    t = np.concatenate([raw[0::2], raw[1::2]]).astype(np.int32)
    out = t.copy()
    out[1:] = (t[1:] - t[:-1] + 128 + 256) & 255
    return out.astype(np.uint8).tobytes()


def _rle(data):
    out, i, n = bytearray(), 0, len(data)
    while i < n:
        j = i
        while j + 1 < n and data[j + 1] == data[i] and j - i < 127:
            j += 1
        if j - i >= 2:                              # a run of 3 .. 128 equal bytes: (count - 1, byte)
            out += bytes([j - i, data[i]])
            i = j + 1
            continue
        k = i                                       # literals until the next run of 3 (at most 127)
        while k < n and k - i < 127 and not (k + 2 < n and data[k] == data[k + 1] == data[k + 2]):
            k += 1
        out += struct.pack("b", -(k - i)) + data[i:k]
        i = k
    return bytes(out)


def _pxr24(rows, types):
    """rows: per scanline, per channel (file order) a numpy array of w samples (f16 -> uint16 view, f32)"""
    out = bytearray()
    for line in rows:
        for samples, ty in zip(line, types):
            if ty == 1:
                v = samples.view(np.uint16).astype(np.uint32)
                d = (np.diff(np.concatenate([[0], v])) & 0xffff).astype(np.uint32)
                out += (d >> 8).astype(np.uint8).tobytes() + (d & 255).astype(np.uint8).tobytes()
            else:
                v = samples.view(np.uint32).astype(np.uint64) & 0xffffff00          # 24 bits survive (the test data has none below)
                d = (np.diff(np.concatenate([[0], v]).astype(np.int64)) & 0xffffffff).astype(np.uint64)
                out += ((d >> 24) & 255).astype(np.uint8).tobytes() + ((d >> 16) & 255).astype(np.uint8).tobytes() + ((d >> 8) & 255).astype(np.uint8).tobytes()
    return zlib.compress(bytes(out))


def write_exr(path, img, compression, half=(False, False, False), extra_alpha=False, origin=(0, 0), decreasing_y=False):
    h, w, _ = img.shape
    names = ["B", "G", "R"] + (["A"] if extra_alpha else [])          # alphabetical, as OpenEXR stores them: A B G R
    names.sort()
    types = {"R": 1 if half[0] else 2, "G": 1 if half[1] else 2, "B": 1 if half[2] else 2, "A": 1}
    planes = {"R": img[..., 0], "G": img[..., 1], "B": img[..., 2], "A": np.ones((h, w), np.float32)}
    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", types[n], 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    x0, y0 = origin
    box = struct.pack("<iiii", x0, y0, x0 + w - 1, y0 + h - 1)
    hdr = struct.pack("<ii", 20000630, 2)
    hdr += _attr("channels", "chlist", chlist) + _attr("compression", "compression", bytes([compression]))
    hdr += _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box)
    hdr += _attr("lineOrder", "lineOrder", bytes([1 if decreasing_y else 0])) + _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += _attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0)) + _attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    per = 16 if compression in (ZIP, PXR24) else 1
    blocks = []
    for b0 in range(0, h, per):
        rows = []
        for y in range(b0, min(h, b0 + per)):
            rows.append([planes[n][y].astype(np.float16 if types[n] == 1 else np.float32) for n in names])
        raw = b"".join(s.tobytes() for line in rows for s in line)
        if compression == NONE:
            data = raw
        elif compression == RLE:
            data = _rle(_predict_and_split(raw))
        elif compression in (ZIPS, ZIP):
            data = zlib.compress(_predict_and_split(raw))
        elif compression == PXR24:
            data = _pxr24(rows, [types[n] for n in names])
        else:
            data = b"\0" * 8
        if len(data) >= len(raw) and compression != PIZ:
            data = raw                                   # OpenEXR stores a block raw when compression does not shrink it
        blocks.append((y0 + b0, data))
    order = list(reversed(blocks)) if decreasing_y else blocks         # file order; the offset table is always by increasing y
    pos = len(hdr) + 8 * len(blocks)
    where = {}
    body = b""
    for y, data in order:
        where[y] = pos + len(body)
        body += struct.pack("<ii", y, len(data)) + data
    table = b"".join(struct.pack("<Q", where[y]) for y, _ in blocks)
    with open(path, "wb") as fh:
        fh.write(hdr + table + body)


def _image(h, w, seed, smooth):
    rng = np.random.default_rng(seed)
    if smooth:          # long runs and small deltas: the compressed blocks really are smaller than the raw ones
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([np.floor(x / 8.0) * 0.25, np.floor(y / 4.0) * 0.5, np.full((h, w), 1.5)], axis=-1)
    else:
        img = rng.uniform(0, 16, size=(h, w, 3))
    # values exactly representable in f16 and in the top 24 bits of an f32: every variant must give them back exactly
    return np.round(img.astype(np.float32) * 64) / np.float32(64)


@pytest.mark.parametrize("compression", [NONE, RLE, ZIPS, ZIP, PXR24])
@pytest.mark.parametrize("smooth", [True, False])
@pytest.mark.parametrize("half", [(False, False, False), (True, True, True), (True, False, True)])
def test_every_supported_compression_reads_back_exactly(tmp_path, compression, smooth, half):
    img = _image(37, 53, compression, smooth)              # 37 rows: a short last block for the 16-line codecs
    p = str(tmp_path / "t.exr")
    write_exr(p, img, compression, half=half, extra_alpha=True, origin=(-5, 7), decreasing_y=(compression == ZIPS))
    got = spt.read_exr(p)
    assert got.shape == img.shape and np.array_equal(got, img)


def test_pxr24_drops_the_low_byte_of_a_float_like_openexr():
    """a 32-bit float written through PXR24 keeps 24 bits: the reader returns exactly those"""
    import tempfile, os
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 1000, size=(20, 31, 3)).astype(np.float32)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "t.exr")
        write_exr(p, img, PXR24)
        got = spt.read_exr(p)
    want = (img.view(np.uint32) & np.uint32(0xffffff00)).view(np.float32)
    assert np.array_equal(got, want)


def test_unsupported_and_corrupt_files_are_reported(tmp_path):
    img = _image(8, 8, 1, True)
    p = str(tmp_path / "piz.exr")
    write_exr(p, img, PIZ)
    with pytest.raises(spt.SptError) as e:
        spt.read_exr(p)
    assert "PIZ" in str(e.value)
    p = str(tmp_path / "rle.exr")
    write_exr(p, _image(16, 64, 2, True), RLE)
    data = bytearray(open(p, "rb").read())
    n_rows = 16
    table_end = data.index(b"screenWindowWidth") + len(b"screenWindowWidth\0float\0") + 4 + 4 + 1 + 8 * n_rows
    first_chunk = struct.unpack_from("<Q", data, table_end - 8 * n_rows)[0]
    assert struct.unpack_from("<i", data, first_chunk)[0] == 0
    data[first_chunk + 8] = 0x7f                              # the first count byte now claims a run of 128: lengths no longer add up
    open(p, "wb").write(bytes(data))
    with pytest.raises(spt.SptError) as e:
        spt.read_exr(p)
    assert "RLE" in str(e.value)
