"""include/spt_detmath.h must return identical bits on gfx950 and x86-64 (the basis of the
bit-exact radiance parity); also checks IEEE division and sqrt, which the kernels rely on."""
import numpy as np
import pytest

import _util
from test_detmath import _eval

pytestmark = pytest.mark.gpu
spt = _util.load_pkg()


def _same(fn, a, b=None):
    cpu = _eval(fn, a, b)
    gpu = spt.device_detmath(fn, a, b)
    bad = np.nonzero(cpu.view(np.uint32) != gpu.view(np.uint32))[0]
    # NaN payloads may differ; compare NaN-ness there
    bad = [k for k in bad if not (np.isnan(cpu[k]) and np.isnan(gpu[k]))]
    assert not bad, (fn, len(bad), a[bad[:4]], cpu[bad[:4]], gpu[bad[:4]])


def test_elementary_functions_bit_identical():
    rng = np.random.default_rng(1)
    ang = np.concatenate([np.linspace(0, 2 * np.pi, 300_001), rng.uniform(-20, 20, 100_000)]).astype(np.float32)
    _same(0, ang)
    _same(1, ang)
    pos = np.concatenate([np.linspace(1e-7, 1, 200_001), np.geomspace(1e-37, 1e37, 100_001), [0.0, 1.0]]).astype(np.float32)
    _same(2, pos)
    _same(3, np.concatenate([np.linspace(-100, 90, 300_001), [0.0]]).astype(np.float32))
    unit = np.concatenate([np.linspace(-1, 1, 300_001), [-1.0, 1.0, 1.5, -2.0]]).astype(np.float32)
    _same(4, unit)
    _same(6, unit)
    y, x = rng.normal(size=300_000).astype(np.float32), rng.normal(size=300_000).astype(np.float32)
    y[:10], x[:10] = 0.0, [-1, 1, 0, -0.0, 2, -3, 0.5, -0.5, 1e-30, -1e-30]
    _same(5, y, x)
    v = rng.uniform(-1e4, 1e4, 200_000).astype(np.float32)
    v[:6] = [0.5, 1.5, -0.5, 2.5, 0.49999997, -2.5]
    _same(7, v)
    _same(8, v)
    # image-texture helpers: pow (sRGB decode), log2 (mip level), trunc / fract (wrap), total floor / round
    base = np.concatenate([np.linspace(0.0, 1.2, 200_001), [0.0, 1.0]]).astype(np.float32)
    _same(13, base, np.full_like(base, 2.4))
    _same(14, np.geomspace(1e-4, 1e5, 100_001).astype(np.float32))
    w = np.concatenate([rng.uniform(-50, 50, 100_000), [np.nan, np.inf, -np.inf, 1e30, -1e30, -0.0, 8388608.0]]).astype(np.float32)
    for fn in (15, 16, 7, 8):
        _same(fn, w)


def test_ieee_sqrt_div_min_max_bit_identical():
    rng = np.random.default_rng(2)
    a = np.concatenate([rng.uniform(0, 1e6, 300_000), np.geomspace(1e-38, 1e38, 50_000), [0.0, np.inf, -1.0]]).astype(np.float32)
    _same(9, a)
    x = rng.normal(size=400_000).astype(np.float32) * np.float32(1e3)
    y = rng.normal(size=400_000).astype(np.float32)
    y[:5] = [0.0, -0.0, np.inf, 1e-38, 1e38]
    _same(10, x, y)
    z = x.copy()
    z[::7] = np.nan
    # max/min: NaN operands are ignored on both sides (values, not zero signs, are compared)
    for fn in (11, 12):
        cpu, gpu = _eval(fn, z, y), spt.device_detmath(fn, z, y)
        assert np.array_equal(np.isnan(cpu), np.isnan(gpu))
        ok = ~np.isnan(cpu)
        assert np.array_equal(cpu[ok], gpu[ok])
