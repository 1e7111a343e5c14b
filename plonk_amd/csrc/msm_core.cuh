// Scalar decomposition shared by the MSM kernels and the host-side window planner.
#pragma once
#include "field.cuh"

namespace plonk {

// Number of signed c-bit windows that cover a 255-bit scalar and absorb the last carry:
// floor(255 / c) + 1 leaves at least one spare bit above bit 254 in the top window.
PLONK_HD constexpr int msm_num_windows(int c) { return 255 / c + 1; }

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
