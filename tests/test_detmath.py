"""include/spt_detmath.h — the deterministic math / RNG / sampler spec shared by kernels and oracle.

Pins its accuracy against numpy's (libm-backed) f32 functions over the argument ranges the path
tracer uses, so that sharing the header cannot hide a wrong sin/cos/log/... from the parity tests.
(The GPU-side evaluation of the same functions is compared bit-for-bit in test_gpu_detmath.py.)
"""
import ctypes as C

import numpy as np

import _util


def _eval(fn, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), dtype=np.float32)
    out = np.zeros_like(a)
    _util.oracle_lib().oracle_detmath(fn, a.size, a.ctypes.data, b.ctypes.data, out.ctypes.data)
    return out


def _ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / np.maximum(ulp, 1e-45)


def test_sin_cos_within_2ulp_abs_on_0_2pi():
    x = np.linspace(0.0, 2.0 * np.pi, 400_001, dtype=np.float32)
    x64 = x.astype(np.float64)
    s, c = _eval(0, x), _eval(1, x)
    # absolute error (values near zero crossings have tiny ulps; directions only need absolute accuracy)
    assert np.abs(s - np.sin(x64)).max() < 2.5e-7
    assert np.abs(c - np.cos(x64)).max() < 2.5e-7
    big = np.abs(np.sin(x64)) > 0.05
    assert _ulp_err(s[big], np.sin(x64[big])).max() <= 2.5
    big = np.abs(np.cos(x64)) > 0.05
    assert _ulp_err(c[big], np.cos(x64[big])).max() <= 2.5


def test_log_exp_within_2ulp():
    x = np.concatenate([np.linspace(1e-7, 1.0, 200_001), np.geomspace(1e-30, 1e30, 50_001)]).astype(np.float32)
    x = x[x > 0]
    assert _ulp_err(_eval(2, x), np.log(x.astype(np.float64))).max() <= 2.0
    y = np.linspace(-80.0, 20.0, 300_001, dtype=np.float32)
    assert _ulp_err(_eval(3, y), np.exp(y.astype(np.float64))).max() <= 2.0
    assert _eval(3, np.array([-200.0], np.float32))[0] == 0.0
    assert np.isinf(_eval(2, np.array([0.0], np.float32))[0]) and _eval(2, np.array([0.0], np.float32))[0] < 0
    assert np.isnan(_eval(2, np.array([-1.0], np.float32))[0])


def test_acos_asin_atan2():
    x = np.linspace(-1.0, 1.0, 200_001, dtype=np.float32)
    assert np.abs(_eval(4, x) - np.arccos(x.astype(np.float64))).max() < 5e-7
    assert np.abs(_eval(6, x) - np.arcsin(x.astype(np.float64))).max() < 3e-7
    assert np.isnan(_eval(4, np.array([1.5], np.float32))[0])
    rng = np.random.default_rng(0)
    y, xx = rng.normal(size=200_000).astype(np.float32), rng.normal(size=200_000).astype(np.float32)
    assert np.abs(_eval(5, y, xx) - np.arctan2(y.astype(np.float64), xx.astype(np.float64))).max() < 6e-7
    # axis cases: atan2(0, -1) = pi, atan2(-0, -1) = -pi, atan2(1, 0) = pi/2
    got = _eval(5, np.array([0.0, -0.0, 1.0, -1.0, 0.0], np.float32), np.array([-1.0, -1.0, 0.0, 0.0, 1.0], np.float32))
    assert np.allclose(got, [np.pi, -np.pi, np.pi / 2, -np.pi / 2, 0.0], atol=1e-6)


def test_round_floor_match_rust_semantics():
    x = np.array([0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 1e6 + 0.5, -2.5, 0.49999997], np.float32)
    # Rust f32::round: half away from zero
    exp = np.sign(x) * np.floor(np.abs(x.astype(np.float64)) + 0.5)
    assert np.array_equal(_eval(7, x), exp.astype(np.float32))
    y = np.array([-1.5, -1.0, -0.0, 0.0, 0.99, 3.0, -7.25], np.float32)
    assert np.array_equal(_eval(8, y), np.floor(y))


def test_pow_log2_trunc_fract_for_textures():
    # sRGB decode: pow(x, 2.4) on ((s + 0.055) / 1.055) for s in (0.04045, 1]
    x = np.linspace(0.09, 1.0, 200_001, dtype=np.float32)
    e = np.full_like(x, 2.4)
    ref = np.power(x.astype(np.float64), np.float64(np.float32(2.4)))
    assert (np.abs(_eval(13, x, e) - ref) / ref).max() < 1.2e-6
    assert _eval(13, np.array([0.0, 0.0, 2.0], np.float32), np.array([2.4, 0.0, 0.0], np.float32)).tolist() == [0.0, 1.0, 1.0]
    # mip level: log2 over footprints 1e-3 .. 1e4 texels
    y = np.geomspace(1e-3, 1e4, 100_001).astype(np.float32)
    assert np.abs(_eval(14, y) - np.log2(y.astype(np.float64))).max() < 2e-6
    # f32::trunc / fract (wrap modes), NaN and huge values pass through floor / round untouched
    z = np.array([1.75, -1.75, 0.25, -0.25, 3.0, -0.0, 8388608.0, 1e30, -1e30], np.float32)
    assert np.array_equal(_eval(15, z), np.trunc(z))
    assert np.array_equal(_eval(16, z), z - np.trunc(z))
    assert np.isnan(_eval(16, np.array([np.inf], np.float32))[0])
    big = np.array([1e30, -1e30, np.inf, -np.inf, 16777216.0], np.float32)
    assert np.array_equal(_eval(8, big), big) and np.array_equal(_eval(7, big), big)
    assert np.isnan(_eval(8, np.array([np.nan], np.float32))[0]) and np.isnan(_eval(7, np.array([np.nan], np.float32))[0])


def test_rng_is_pcg32_xsh_rr_and_uniform():
    lib = _util.oracle_lib()
    # independent re-implementation of the stream definition
    def splitmix(z):
        z = (z + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        return z ^ (z >> 31)
    seed, pixel, sample = 12345, 777, 42
    st = splitmix(splitmix(seed) ^ ((pixel << 32) | sample))
    assert lib.oracle_rng_state(seed, pixel, sample) == st
    exp = []
    for _ in range(16):
        old = st
        st = (old * 6364136223846793005 + 1442695040888963407) & (2**64 - 1)
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        out = ((xs >> rot) | (xs << ((32 - rot) & 31))) & 0xFFFFFFFF
        exp.append(np.float32(out >> 8) * np.float32(2.0 ** -24))
    got = np.zeros(16, np.float32)
    lib.oracle_rng_stream(seed, pixel, sample, 16, got.ctypes.data)
    assert np.array_equal(got, np.array(exp, np.float32))
    big = np.zeros(200_000, np.float32)
    lib.oracle_rng_stream(1, 0, 0, big.size, big.ctypes.data)
    assert 0.0 <= big.min() and big.max() < 1.0
    assert abs(big.mean() - 0.5) < 3e-3 and abs(big.var() - 1 / 12) < 2e-3


def test_r2_closed_form_tracks_the_reference_recurrence():
    """recurrence.rs:44-52 accumulates in f32; the closed form is its exact-arithmetic counterpart."""
    spp = 256
    got = np.zeros((spp, 2), np.float32)
    _util.oracle_lib().oracle_r2_offsets(0, spp, spp, got.ctypes.data)
    a = np.float32(0.754877666246571)
    a2 = np.float32(a * a)
    x, y = np.float32(0.5), np.float32(0.5)
    ref = []
    for _ in range(spp):
        x = np.float32(x + a)
        if x >= 1.0:
            x = np.float32(x - 1.0)
        y = np.float32(y + a2)
        if y >= 1.0:
            y = np.float32(y - 1.0)
        ref.append((x, y))
    ref = np.array(ref, np.float32)
    assert (got >= 0).all() and (got < 1).all()
    assert np.abs(got - ref).max() < 2e-5  # f32 accumulation drift of the reference over 256 steps
    # pixel p continues the sequence where pixel p-1 stopped (state carried across pixels, quirk Q8)
    nxt = np.zeros((1, 2), np.float32)
    _util.oracle_lib().oracle_r2_offsets(1, spp, 1, nxt.ctypes.data)
    step = (nxt[0] - got[-1]) % 1.0
    assert np.allclose(step, [float(a), float(a2)], atol=1e-6)
