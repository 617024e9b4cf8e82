"""Output packing + LPM tonemap: the step right after the path.  CPU tests of the oracle restatement and of
libart's host-side LpmSetup port (no GPU needed); the GPU comparison lives in test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest


def _values(n, seed):
    k = np.arange(n, dtype=np.float64)
    x = np.sin(k * 12.9898 + seed) * 43758.5453
    u = x - np.floor(x)
    return (np.exp((u - 0.5) * 40.0)).astype(np.float32)          # 2e-9 .. 5e8, log-uniform


def test_f16_pack_matches_numpy(orc):
    vals = np.concatenate([_values(4000, 1), -_values(500, 2), np.array([0.0, -0.0, 65504.0, 65520.0, 1e-8, 6.1e-5, 5.96e-8, 2.98e-8, np.inf, -np.inf], np.float32)])
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    got = np.array([orc.pack_f16(float(v)) for v in vals], np.uint16)
    assert np.array_equal(got, want)


def test_b10g11r11_is_the_nearest_representable(orc):
    """B10G11R11_UFLOAT_PACK32 (renderer.rs:268): every channel rounds to the nearest 11/10-bit unsigned float, ties to even"""
    table11 = np.array([orc.unpack_b10g11r11(v)[0] for v in range(0x7C0)], np.float64)           # all finite 11-bit values
    table10 = np.array([orc.unpack_b10g11r11(v << 22)[2] for v in range(0x3E0)], np.float64)
    assert np.all(np.diff(table11) > 0) and np.all(np.diff(table10) > 0) and table11[0] == 0 and table11[-1] == 65024.0 and table10[-1] == 64512.0
    for v in _values(1500, 3):
        if v >= 65024.0:
            continue
        pk = orc.pack_b10g11r11([v, v, v])
        r, g, b = pk & 0x7FF, (pk >> 11) & 0x7FF, pk >> 22
        assert r == g
        for code, table in ((r, table11), (b, table10)):
            d = np.abs(table - float(v))
            best = int(np.argmin(d))
            assert d[code] == d[best], (v, code, best)
            if best + 1 < len(table) and d[best] == d[best + 1]:
                assert code % 2 == 0                                                                  # tie -> even mantissa
    assert orc.pack_b10g11r11([-1.0, 0.0, 1.0]) == (0 | (0 << 11) | (0x1E0 << 22)) and np.array_equal(orc.unpack_b10g11r11(0x681c03c0), [1.0, 0.5, 0.25])


def test_lpm_control_block_port_matches_oracle_and_closed_forms(orc):
    """LpmData::new(false, 0, 256, 8, 0.25, 1, 0, (1, 1/2, 1/32)) of vk_tonemap.rs:417-426"""
    from araytracingjourney_amd import _lib
    ctl = np.zeros(96, np.uint32)
    sat, ct = (C.c_float * 3)(0, 0, 0), (C.c_float * 3)(1.0, 0.5, 1.0 / 32.0)
    assert _lib.load().art_lpm_control_block(0, 0.0, 256.0, 8.0, 0.25, 1.0, sat, ct, ctl.ctypes.data_as(C.c_void_p)) == 0
    ref = orc.lpm_control_block(0, 0.0, 256.0, 8.0, 0.25, 1.0, (0, 0, 0), (1.0, 0.5, 1.0 / 32.0))
    a, b = ctl.view(np.float32), ref.view(np.float32)
    assert np.allclose(a, b, rtol=1e-6, atol=0)
    f = a
    assert np.allclose(f[0:4], 1.25) and np.allclose(f[9:12], [1.0, 0.5, 1 / 32])
    assert np.allclose(f[6:9], [0.2126, 0.7152, 0.0722], atol=2e-4) and np.allclose(f[12:15], 1.0 / f[6:9], rtol=1e-6)   # Rec.709 luma
    # tone curve: y = x^c / (x^c * a + b) must map midIn -> 0.18 and hdrMax -> 1 (what toneScaleBias is solved for)
    c_, a_, b_ = 1.25, float(f[4]), float(f[5])
    mid_in = 256.0 * 0.18 * 2.0 ** -8
    curve = lambda x: x ** c_ / (x ** c_ * a_ + b_)
    assert abs(curve(mid_in) - 0.18) < 1e-4 and abs(curve(256.0) - 1.0) < 1e-4
    assert np.all(ctl[40:] == 0)


def test_present_properties(orc):
    grey = np.zeros((1, 6, 4), np.float32)
    grey[0, :, :3] = np.array([0.0, 0.02, 0.18, 1.0, 16.0, 256.0], np.float32)[:, None]
    packed, bgra = orc.present(grey)
    assert np.all(bgra[0, :, 0] == bgra[0, :, 1]) and np.all(bgra[0, :, 1] == bgra[0, :, 2]) and np.all(bgra[..., 3] == 255)
    assert bgra[0, 0, 0] == 0 and np.all(np.diff(bgra[0, :, 0].astype(int)) > 0) and bgra[0, -1, 0] == 255
    assert abs(int(bgra[0, 2, 0]) - round(0.18 ** (1 / 2.2) * 255)) <= 1                       # 18 % grey stays 18 % grey (exposure 8 stops under 256)
    half = orc.present(grey, np.full((1, 6), 128, np.uint32))[1]
    assert np.all(half[0, 1:5, 0] < bgra[0, 1:5, 0]) and half[0, 5, 0] <= bgra[0, 5, 0]                                              # ao/255 darkens (tonemap.comp.glsl:33-34)
    red = np.zeros((1, 1, 4), np.float32); red[0, 0, 0] = 500.0
    out = orc.present(red)[1][0, 0]
    assert out[2] == 255 and out[1] > 0 and out[0] > 0 and out[1] > out[0]                     # over-exposed red bleeds by the crosstalk (1, 1/2, 1/32)
