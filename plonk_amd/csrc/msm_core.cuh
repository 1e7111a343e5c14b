// Scalar decomposition shared by the MSM kernels and the host-side window planner.
#pragma once
#include "field.cuh"

namespace plonk {

// Number of signed c-bit windows: ceil(255 / c).  Where c divides 255 (15, 17) this relies on the digits being
// taken of k or of r - k, whichever is the smaller (scalar_fold: at most (r - 1) / 2 < 2^254): the top window,
// which ends at bit 254, then holds less than 2^(c-1) and its digit never carries out.  (Without the fold
// such a width needs one more window for that carry, with half of all entries in a single bucket.)  For every
// other width the windows reach beyond bit 254 anyway and the scalar is used as it is: folding would halve
// the range of the narrow top window, i.e. pile its digits into half as many buckets.
PLONK_HD constexpr int msm_num_windows(int c) { return (255 + c - 1) / c; }
PLONK_HD constexpr bool msm_folds_scalars(int c) { return 255 % c == 0; }

// k (canonical, not in Montgomery form) becomes min(k, r - k); true when it was replaced, in which case the
// caller uses the opposite point: k P = (r - k)(-P).
PLONK_HD bool scalar_fold(Fr &k) {
    const Fr n = fe_neg(k);  // r - k, and 0 for k = 0
    bool less = false, decided = false;
#pragma unroll
    for (int i = FrCfg::N - 1; i >= 0; i--) {
        if (!decided && n.l[i] != k.l[i]) {
            less = n.l[i] < k.l[i];
            decided = true;
        }
    }
    if (less) k = n;
    return less;
}

// Signed digit of window w of the canonical (non-Montgomery) scalar k.
//   raw = bits [w*c, w*c + c) of k, d = raw + carry_in;
//   d > 2^(c-1)  ->  d -= 2^c, carry_out = 1.
// Result lies in [-(2^(c-1) - 1), 2^(c-1)]: |d| - 1 indexes 2^(c-1) buckets, d == 0 is skipped.
// Requires 2 <= c <= 30.
PLONK_HD int32_t signed_digit(const Fr &k, int w, int c, uint32_t &carry) {
    const int bit = w * c;
    const int idx = bit >> 5, sh = bit & 31;
    // select chain instead of k.l[idx]: a run-time index would send k to scratch memory
    uint32_t lo32 = 0, hi32 = 0;
#pragma unroll
    for (int i = 0; i < FrCfg::N; i++) {
        lo32 = (idx == i) ? k.l[i] : lo32;
        hi32 = (idx + 1 == i) ? k.l[i] : hi32;
    }
    uint64_t lo = lo32, hi = hi32;
    uint64_t win = ((lo | (hi << 32)) >> sh) & ((1ull << c) - 1);
    int64_t d = (int64_t)win + carry;
    carry = 0;
    if (d > (int64_t)(1ll << (c - 1))) {
        d -= (1ll << c);
        carry = 1;
    }
    return (int32_t)d;
}

}  // namespace plonk
