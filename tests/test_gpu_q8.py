"""-m gpu: the 8-bit surfaces' gamma shortcut (rd_kernels.h::rd_q8_gamma) is the pinned function, code for code.

  * on the device, for ALL 2^32 float encodings: rd_q8_gamma(x) == trunc(255 * rd_gamma_clamp(x) + 0.5) (rd_selftest_q8);
  * against the ORACLE (ref_powf -> clamp -> ref_pack_u8): every encoding in a window around each of the 255 code
    steps, the special encodings, and a stride through all 2^32 (tools/q8_exhaustive.hip does every encoding against the
    oracle; its output is kept in profiles/).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _codes(ra, first, n, lut=False):
    from raweditor_amd import _lib
    out = np.empty(n, np.uint8)
    fn = _lib.lib().rd_selftest_q8_lut_codes if lut else _lib.lib().rd_selftest_q8_codes
    _lib.check(fn(0, first, n, out.ctypes.data_as(C.c_void_p)))
    return out


def _oracle_codes(refc, bits):
    L = refc.lib()
    x = bits.astype(np.uint32).view(np.float32)
    g = np.empty_like(x)
    for i, v in enumerate(x):                        # scalar ref_powf through ctypes: keep the samples modest
        g[i] = L.ref_powf(C.c_float(v), C.c_float(np.float32(0.45454547)), 0)
    g = np.where(g > 0, g, np.float32(0)).astype(np.float32)      # max(c, 0), NaN -> 0
    g = np.minimum(g, np.float32(1))
    return refc.pack_u8(g)


def test_q8_shortcut_equals_the_pinned_codes_for_every_float(gpu_lib):
    from raweditor_amd import _lib
    bad, first, fb, dist = C.c_uint64(), C.c_uint32(), C.c_uint64(), C.c_float()
    _lib.check(_lib.lib().rd_selftest_q8(0, C.byref(bad), C.byref(first), C.byref(fb), C.byref(dist)))
    assert bad.value == 0, f"{bad.value} encodings differ, first 0x{first.value:08x}"
    assert dist.value * 4 < 0.00025, dist.value      # RD_Q8_EPS = 2.5e-4: at least 4x the largest distance seen here
    assert 0 < fb.value < 2**31 * 0.01               # the pinned evaluation is the rare path


def test_q8_threshold_table_equals_the_pinned_codes_for_every_float(gpu_lib):
    """Round 4: the export kernel's RGBA8 / RGB8 codes come from a threshold table in LDS (rd_q8_lut_bits): for ALL 2^32 float
    encodings -- negative, NaN, denormal, > 1 included -- the table's code is trunc(255 * rd_gamma_clamp(x) + 0.5)."""
    from raweditor_amd import _lib
    bad, first = C.c_uint64(), C.c_uint32()
    _lib.check(_lib.lib().rd_selftest_q8_lut(0, C.byref(bad), C.byref(first)))
    assert bad.value == 0, f"{bad.value} encodings differ, first 0x{first.value:08x}"


@pytest.mark.parametrize("lut", [False, True], ids=["transcendental-shortcut", "threshold-table"])
def test_q8_shortcut_against_the_oracle(gpu_lib, refc, lut):
    ra = gpu_lib
    # where the oracle's code changes: bisect each of the 255 steps over the non-negative encodings
    L = refc.lib()

    def oracle_code(bits):
        return int(_oracle_codes(refc, np.array([bits], np.uint32))[0])

    assert oracle_code(0) == 0 and oracle_code(0x3f800000) == 255
    checked = 0
    for k in range(1, 256):
        lo, hi = 0, 0x3f800000                       # code(lo) < k <= code(hi)
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if oracle_code(mid) >= k:
                hi = mid
            else:
                lo = mid
        first = (max(hi - 512, 0) // 256) * 256
        bits = np.arange(first, first + 1024, dtype=np.uint32)
        assert np.array_equal(_codes(ra, first, 1024, lut), _oracle_codes(refc, bits)), f"step to code {k} near 0x{hi:08x}"
        checked += 1024
    # specials and a stride through every exponent, both signs
    for first in (0x00000000, 0x007fff00, 0x00800000, 0x3f7fff00, 0x3f800000, 0x7f7fff00, 0x7f800000, 0x7fc00000,
                  0x80000000, 0xbf800000, 0xff800000, 0xffffff00):
        bits = np.arange(first, first + 256, dtype=np.uint64).astype(np.uint32)
        assert np.array_equal(_codes(ra, first, 256, lut), _oracle_codes(refc, bits)), hex(first)
    rng = np.random.default_rng(8)
    for first in rng.integers(0, 2**32 - 256, 64, dtype=np.uint64):
        first = int(first) // 256 * 256
        bits = np.arange(first, first + 256, dtype=np.uint64).astype(np.uint32)
        assert np.array_equal(_codes(ra, first, 256, lut), _oracle_codes(refc, bits)), hex(first)
    assert checked == 255 * 1024


def test_f16_shortcut_equals_binary16_of_the_pinned_gamma_for_every_float(gpu_lib):
    """rd_f16_gamma (the RGBA-f16 surface, BASELINE config 5): for all 2^32 encodings the half is binary16(rd_gamma_clamp(x))
    and the histogram code is the pinned one, with and without the fused histogram."""
    from raweditor_amd import _lib
    bad, first, fb = C.c_uint64(), C.c_uint32(), C.c_uint64()
    _lib.check(_lib.lib().rd_selftest_f16(0, C.byref(bad), C.byref(first), C.byref(fb)))
    assert bad.value == 0, f"{bad.value} encodings differ, first 0x{first.value:08x}"
    assert 0 < fb.value < 31 * 2**23 * 0.02           # of the ~31 binades of x whose gamma is a normal half: a rare path


def test_f16_shortcut_against_the_oracle(gpu_lib, refc):
    from raweditor_amd import _lib

    def halves(first, n):
        out = np.empty(n, np.uint16)
        _lib.check(_lib.lib().rd_selftest_f16_halves(0, first, n, out.ctypes.data_as(C.c_void_p)))
        return out

    def oracle_halves(bits):
        L = refc.lib()
        x = bits.astype(np.uint32).view(np.float32)
        g = np.empty_like(x)
        for i, v in enumerate(x):
            g[i] = L.ref_powf(C.c_float(v), C.c_float(np.float32(0.45454547)), 0)
        g = np.minimum(np.where(g > 0, g, np.float32(0)).astype(np.float32), np.float32(1))
        return refc.pack_f16(g).view(np.uint16)

    rng = np.random.default_rng(16)
    firsts = [0x00000000, 0x007fff00, 0x00800000, 0x3f7fff00, 0x3f800000, 0x7f7fff00, 0x7f800000, 0x7fc00000, 0x80000000,
              0xbf800000, 0xff800000, 0xffffff00]
    # the bright half of the range, where almost all pixels live, densely; then every exponent; then anywhere
    firsts += [int(v) // 256 * 256 for v in rng.integers(0x3c000000, 0x3f800000, 160, dtype=np.uint64)]
    firsts += [(e << 23) + (int(m) // 256 * 256) for e in range(1, 255) for m in rng.integers(0, 1 << 23, 1)]
    firsts += [int(v) // 256 * 256 for v in rng.integers(0, 2**32 - 256, 64, dtype=np.uint64)]
    for first in firsts:
        bits = np.arange(first, first + 256, dtype=np.uint64).astype(np.uint32)
        assert np.array_equal(halves(first, 256), oracle_halves(bits)), hex(first)


def test_f16_tables_equal_binary16_of_the_pinned_gamma_for_every_float(gpu_lib):
    """Round 4: the export kernel's RGBA-f16 surface takes halves AND histogram codes from two-level threshold tables in LDS
    (rd_f16_lut_lookup); the lanes it sends to the pinned evaluation are 0 < x < 2^-16 and the one non-monotone encoding.  For
    all 2^32 encodings: half == binary16(rd_gamma_clamp(x)), code == the pinned code, no stray high bits."""
    from raweditor_amd import _lib
    bad, first, pinned = C.c_uint64(), C.c_uint32(), C.c_uint64()
    _lib.check(_lib.lib().rd_selftest_f16_lut(0, C.byref(bad), C.byref(first), C.byref(pinned)))
    assert bad.value == 0, f"{bad.value} encodings differ, first 0x{first.value:08x}"
    assert pinned.value == 0x37800000 - 1 + 1          # the encodings of (0, 2^-16), and the dip


def test_f16_tables_against_the_oracle(gpu_lib, refc):
    """The same lookup against the ORACLE's pow -> clamp -> binary16 / 8-bit pack: the special encodings, the dip and its
    neighbours, the edge of the tables' domain (2^-16), the bright half densely, every exponent, and windows around steps."""
    from raweditor_amd import _lib

    def values(first, n):
        out = np.empty(n, np.uint32)
        _lib.check(_lib.lib().rd_selftest_f16_lut_values(0, first, n, out.ctypes.data_as(C.c_void_p)))
        return out

    def oracle(bits):
        L = refc.lib()
        x = bits.astype(np.uint32).view(np.float32)
        g = np.empty_like(x)
        for i, v in enumerate(x):
            g[i] = L.ref_powf(C.c_float(v), C.c_float(np.float32(0.45454547)), 0)
        g = np.minimum(np.where(g > 0, g, np.float32(0)).astype(np.float32), np.float32(1))
        return refc.pack_f16(g).view(np.uint16).astype(np.uint32) | (refc.pack_u8(g).astype(np.uint32) << 16)

    rng = np.random.default_rng(17)
    firsts = [0x00000000, 0x007fff00, 0x00800000, 0x377fff00, 0x37800000, 0x3eefb500, 0x3f7fff00, 0x3f800000, 0x7f7fff00,
              0x7f800000, 0x7fc00000, 0x80000000, 0xbf800000, 0xff800000, 0xffffff00]
    firsts += [int(v) // 256 * 256 for v in rng.integers(0x3c000000, 0x3f800000, 160, dtype=np.uint64)]
    firsts += [(e << 23) + (int(m) // 256 * 256) for e in range(1, 255) for m in rng.integers(0, 1 << 23, 1)]
    firsts += [int(v) // 256 * 256 for v in rng.integers(0, 2**32 - 256, 64, dtype=np.uint64)]
    # windows on fine-bucket boundaries (2^13 encodings) inside the domain: where one table entry hands over to the next
    firsts += [((int(v) >> 13) << 13) - 128 for v in rng.integers(0x37800000 + 8192, 0x3f800000, 96, dtype=np.uint64)]
    for first in firsts:
        bits = np.arange(first, first + 256, dtype=np.uint64).astype(np.uint32)
        assert np.array_equal(values(first, 256), oracle(bits)), hex(first)
