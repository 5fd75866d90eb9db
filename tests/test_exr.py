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


# ---- PIZ (ImfPizCompressor / ImfHuf / ImfWav, written from the format description like the decoder it tests) ----------
def _wenc14(a, b):
    a = a - 65536 if a >= 32768 else a
    b = b - 65536 if b >= 32768 else b
    return ((a + b) >> 1) & 0xffff, (a - b) & 0xffff


def _wenc16(a, b):
    ao = (a + 0x8000) & 0xffff
    m = (ao + b) >> 1
    d = ao - b
    if d < 0:
        m = (m + 0x8000) & 0xffff
    return m, d & 0xffff


def _wav2_encode(buf, start, nx, ox, ny, oy, mx):
    enc = _wenc14 if mx < (1 << 14) else _wenc16
    n = min(nx, ny)
    p, p2 = 1, 2
    while p2 <= n:
        oy1, oy2, ox1, ox2 = oy * p, oy * p2, ox * p, ox * p2
        py, ey = 0, oy * (ny - p2)
        while py <= ey:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                i = start + px
                i00, i01 = enc(buf[i], buf[i + ox1])
                i10, i11 = enc(buf[i + oy1], buf[i + oy1 + ox1])
                buf[i], buf[i + oy1] = enc(i00, i10)
                buf[i + ox1], buf[i + oy1 + ox1] = enc(i01, i11)
                px += ox2
            if nx & p:
                i = start + px
                buf[i], buf[i + oy1] = enc(buf[i], buf[i + oy1])
            py += oy2
        if ny & p:
            px, ex = py, py + ox * (nx - p2)
            while px <= ex:
                i = start + px
                buf[i], buf[i + ox1] = enc(buf[i], buf[i + ox1])
                px += ox2
        p, p2 = p2, p2 << 1


def _huf_compress(data, use_runs=True):
    import heapq
    freq = {}
    for v in data:
        freq[v] = freq.get(v, 0) + 1
    im, i_max = min(freq), max(freq) + 1
    freq[i_max] = 1                                        # the run-length escape
    syms = sorted(freq)
    heap = [(freq[sym], k) for k, sym in enumerate(syms)]
    heapq.heapify(heap)
    parent = list(range(len(syms)))                        # leaves 0 .. n-1, inner nodes appended
    while len(heap) > 1:
        fa, a = heapq.heappop(heap)
        fb, b = heapq.heappop(heap)
        parent.append(len(parent))
        parent[a] = parent[b] = len(parent) - 1
        heapq.heappush(heap, (fa + fb, len(parent) - 1))
    depth = [0] * len(parent)
    for node in range(len(parent) - 2, -1, -1):            # a parent always has the larger index
        depth[node] = depth[parent[node]] + 1
    length = {sym: depth[k] for k, sym in enumerate(syms)}
    assert max(length.values()) <= 58
    # canonical codes exactly as the decoder assigns them
    n = [0] * 59
    for l in length.values():
        n[l] += 1
    n[0] += 65537 - len(length)
    c = 0
    for i in range(58, 0, -1):
        nc = (c + n[i]) >> 1
        n[i] = c
        c = nc
    code = {}
    for sym in range(im, i_max + 1):
        l = length.get(sym, 0)
        if l:
            code[sym] = n[l]
            n[l] += 1
    bits = []                                              # (value, nbits)
    sym = im
    while sym <= i_max:
        l = length.get(sym, 0)
        if l == 0:
            run = 1
            while sym + run <= i_max and length.get(sym + run, 0) == 0 and run < 261:
                run += 1
            if run >= 6:
                bits += [(63, 6), (run - 6, 8)]
                sym += run
                continue
            if run >= 2:
                bits.append((59 + run - 2, 6))
                sym += run
                continue
        bits.append((l, 6))
        sym += 1
    table = _pack_bits(bits)
    out = []

    def send(sym, run):
        ls, lr = length[sym], length[i_max]
        if use_runs and ls + lr + 8 < ls * run:
            out.extend([(code[sym], ls), (code[i_max], lr), (run, 8)])
        else:
            out.extend([(code[sym], ls)] * (run + 1))
    cur, run = data[0], 0
    for v in data[1:]:
        if v == cur and run < 255:
            run += 1
        else:
            send(cur, run)
            cur, run = v, 0
    send(cur, run)
    n_bits = sum(b for _, b in out)
    body = _pack_bits(out)
    return struct.pack("<IIIII", im, i_max, len(table), n_bits, 0) + table + body


def _pack_bits(items):
    acc, nb, out = 0, 0, bytearray()
    for v, b in items:
        acc = (acc << b) | v
        nb += b
        while nb >= 8:
            nb -= 8
            out.append((acc >> nb) & 255)
        acc &= (1 << nb) - 1
    if nb:
        out.append((acc << (8 - nb)) & 255)
    return bytes(out)


def _piz(rows, types, use_runs=True):
    """rows: per scanline, per channel (file order) a numpy array of w samples -> one PIZ block"""
    ny, nx = len(rows), len(rows[0][0])
    buf, starts = [], []
    for c, ty in enumerate(types):
        starts.append(len(buf))
        for line in rows:
            buf += line[c].view(np.uint16).tolist()       # f16: one word per sample; f32: low word, high word
    used = sorted(set(buf) | {0})
    nonzero = [v for v in used if v]
    bitmap = bytearray(8192)
    for v in nonzero:
        bitmap[v >> 3] |= 1 << (v & 7)
    fwd = {v: k for k, v in enumerate(used)}
    buf = [fwd[v] for v in buf]
    mx = len(used) - 1
    for c, ty in enumerate(types):
        wpp = 1 if ty == 1 else 2
        for j in range(wpp):
            _wav2_encode(buf, starts[c] + j, nx, wpp, ny, nx * wpp, mx)
    if nonzero:
        lo, hi = min(nonzero) >> 3, max(nonzero) >> 3
        head = struct.pack("<HH", lo, hi) + bytes(bitmap[lo:hi + 1])
    else:
        head = struct.pack("<HH", 8191, 0)                 # minNonZero > maxNonZero: no bitmap bytes follow
    huf = _huf_compress(buf, use_runs)
    return head + struct.pack("<i", len(huf)) + huf


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
    per = 16 if compression in (ZIP, PXR24) else (32 if compression == PIZ else 1)
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
        elif compression == PIZ:
            data = _piz(rows, [types[n] for n in names])
        else:
            data = b"\0" * 8                             # B44 and the rest: the reader refuses the header
        if len(data) >= len(raw):
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


@pytest.mark.parametrize("compression", [NONE, RLE, ZIPS, ZIP, PIZ, PXR24])
@pytest.mark.parametrize("smooth", [True, False])
@pytest.mark.parametrize("half", [(False, False, False), (True, True, True), (True, False, True)])
def test_every_supported_compression_reads_back_exactly(tmp_path, compression, smooth, half):
    img = _image(37, 53, compression, smooth)              # 37 rows: a short last block for the 16-line codecs
    p = str(tmp_path / "t.exr")
    write_exr(p, img, compression, half=half, extra_alpha=True, origin=(-5, 7), decreasing_y=(compression == ZIPS))
    got = spt.read_exr(p)
    assert got.shape == img.shape and np.array_equal(got, img)


def test_piz_sixteen_bit_wavelet_long_codes_and_runs(tmp_path):
    """A block with more than 2^14 distinct 16-bit words takes the modulo-2^16 wavelet and Huffman codes longer than the
    14-bit decoding table; constant regions take the run-length escape; a block that PIZ cannot shrink is stored raw."""
    rng = np.random.default_rng(11)
    h, w = 40, 700                                          # 32 + 8 lines: 134 k words in the first block
    img = rng.uniform(0, 1000, size=(h, w, 3)).astype(np.float32)
    img[5:20, 100:400] = np.float32(2.5)                    # long runs inside the noise
    p = str(tmp_path / "big.exr")
    write_exr(p, img, PIZ)
    assert np.array_equal(spt.read_exr(p), img)
    flat = np.full((33, 64, 3), np.float32(0.75))           # one value: bitmap of one bit, every code a run
    flat[..., 1] = 0.0                                      # and a channel that is all zero
    write_exr(p, flat, PIZ, half=(True, False, True))
    assert np.array_equal(spt.read_exr(p), flat)
    zero = np.zeros((7, 9, 3), np.float32)                  # nothing but zeros: minNonZero > maxNonZero, no bitmap bytes
    write_exr(p, zero, PIZ)
    assert np.array_equal(spt.read_exr(p), zero)


def test_corrupt_piz_blocks_are_reported_not_followed(tmp_path):
    img = _image(37, 53, 4, True)
    p = str(tmp_path / "t.exr")
    write_exr(p, img, PIZ)
    good = bytearray(open(p, "rb").read())
    assert np.array_equal(spt.read_exr(p), img)
    first = struct.unpack_from("<Q", good, good.index(b"screenWindowWidth") + len(b"screenWindowWidth\0float\0") + 4 + 4 + 1)[0]
    size = struct.unpack_from("<i", good, first + 4)[0]
    rng = np.random.default_rng(0)
    outcomes = set()
    for trial in range(300):
        data = bytearray(good)
        for _ in range(1 + trial % 3):
            data[first + 8 + int(rng.integers(size))] ^= 1 << int(rng.integers(8))
        open(p, "wb").write(bytes(data))
        try:
            got = spt.read_exr(p)
            outcomes.add("read")                            # a flipped data bit can still decode, to other values
            assert got.shape == img.shape
        except spt.SptError as e:
            assert e.status == 101 and "PIZ" in str(e), str(e)
            outcomes.add("refused")
    assert outcomes == {"read", "refused"}
    open(p, "wb").write(bytes(good[:first + 8 + size // 2]))     # truncated inside the block
    with pytest.raises(spt.SptError):
        spt.read_exr(p)


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
    p = str(tmp_path / "b44.exr")
    write_exr(p, img, 6)
    with pytest.raises(spt.SptError) as e:
        spt.read_exr(p)
    assert "B44" in str(e.value) and e.value.status == 103
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
