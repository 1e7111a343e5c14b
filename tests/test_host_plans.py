"""Host-side planning logic of the library (no GPU needed): window plans of the MSM and pass
structure of the transforms."""
import ctypes

import pytest

from plonk_amd import lib as L


@pytest.fixture(scope="module")
def lib():
    return L.load_library()


def msm_plan(lib, n, override=0):
    c, w, b = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
    assert lib.plonk_hip_msm_plan_for(n, override, ctypes.byref(c), ctypes.byref(w), ctypes.byref(b)) == 0
    return c.value, w.value, b.value


def test_msm_plans_cover_a_255_bit_scalar(lib):
    for log_n in list(range(0, 27)):
        for n in {1 << log_n, (1 << log_n) + 1, 3 << max(log_n - 1, 0)}:
            c, nwin, nb = msm_plan(lib, n)
            assert 2 <= c <= 24
            assert nb == 1 << (c - 1)            # signed digits: magnitudes 1 .. 2^(c-1)
            assert nwin == -(-255 // c)           # windows reach bit 254; widths that divide 255 fold the scalar to min(k, r - k) < 2^254
            assert n * nwin < 1 << 31            # table index + sign bit fit one 32-bit word
    for c in (2, 3, 8, 13, 16, 24):
        assert msm_plan(lib, 1000, c)[0] == c   # explicit override wins


def test_large_plans_avoid_narrow_top_windows(lib):
    # 255 mod c small would pile the top window's digits into a few buckets
    for log_n in range(11, 27):
        c, nwin, _ = msm_plan(lib, 1 << log_n)
        assert 255 - (nwin - 1) * c >= 8


@pytest.mark.parametrize("natural", [0, 1])
def test_ntt_pass_plans_partition_the_stages(lib, natural):
    s0 = (ctypes.c_uint32 * 8)()
    s1 = (ctypes.c_uint32 * 8)()
    for log_n in range(3, 31):
        npass = lib.plonk_hip_ntt_plan(log_n, natural, 8, s0, s1)
        assert 1 <= npass <= 4
        assert s0[0] == 0 and s1[npass - 1] == log_n
        for i in range(npass):
            k = s1[i] - s0[i]
            assert 3 <= k <= 11                       # register rounds need 8 elements; LDS holds 2^11
            if i:
                assert s0[i] == s1[i - 1]
        if natural and 11 < log_n <= 20:
            assert npass == 2 and max(s1[0] - s0[0], s1[1] - s0[1]) <= 10   # both sides keep 64-byte rows
    assert lib.plonk_hip_ntt_plan(2, natural, 8, s0, s1) == 0
