// XYZZ addition and doubling cut into stages that four waves work on side by side.
//
// The sums behind the bucket accumulation end in chains of dependent point additions executed by a
// handful of waves while the rest of the GPU idles, and a wave that runs alone on its SIMD issues a
// multiply-add only every other slot.  One addition is 14 field products of which at most four
// depend on each other in a row, so the four waves of a block (one per SIMD of a compute unit) take
// one product each per stage and exchange the results through LDS: 4 product times and 5 barriers
// instead of 14 product times (doubling: 3 instead of 9).
//
// The stage functions are written against a memory accessor M, so that the same code runs on the
// host (one slot, plain arrays) for checking and on the device (64 slots in LDS):
//     Fq28 a(int i), b(int i), t(int i);      coordinate i (0 x, 1 y, 2 zz, 3 zzz) / temporary i
//     void set_a(int i, x), set_t(int i, x);  bool slow(); void set_slow(bool);
// a and b may be the same storage read at different slots; b is never written.  All values are in
// the lazy form of g1_28.cuh (x < 5.1 q, y < 3.1 q, zz, zzz < 1.1 q) and stay in it.
#pragma once
#include "g1_28.cuh"

namespace plonk {

static constexpr int kCoopWaves = 4;
static constexpr int kCoopAddStages = 5;
static constexpr int kCoopDblStages = 4;
static constexpr int kCoopTemps = 8;

// K * a without reduction (K small): limbs stay normalised.
template <int K>
PLONK_HD Fq28 q28_scale_lazy(const Fq28 &a) {
    Fq28 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < Q28::N; i++) {
        uint32_t t = a.l[i] * (uint32_t)K + c;
        r.l[i] = t & Q28::MASK;
        c = t >> Q28::BITS;
    }
    return r;
}

// a <- a + b for two points that are not the identity.  Stage `stage` of wave `w`; a barrier
// separates the stages.  After stage 1 slow() tells that the points are equal or opposite: the later
// stages then leave a alone and the caller resolves the slot on canonical values.
//   temporaries: 0 U1 -> Q, 1 U2 -> P -> PPP, 2 S1, 3 S2 -> R, 4 PP -> R (Q - X3), 5 RR,
//                6 ZZ1 ZZ2 -> S1 PPP, 7 ZZZ1 ZZZ2
template <class M>
PLONK_HD void coop_add_stage(M &m, int stage, int w) {
    if (stage == 0) {
        // U1 = X1 ZZ2, U2 = X2 ZZ1, S1 = Y1 ZZZ2, S2 = Y2 ZZZ1
        const int ia = w == 0 ? 0 : (w == 1 ? 2 : (w == 2 ? 1 : 3));
        const int ib = w == 0 ? 2 : (w == 1 ? 0 : (w == 2 ? 3 : 1));
        m.set_t(w, q28_mul_lazy(m.a(ia), m.b(ib)));
    } else if (stage == 1) {
        // P = U2 - U1, PP = P^2 | R = S2 - S1, RR = R^2 | ZZ1 ZZ2 | ZZZ1 ZZZ2
        Fq28 x, y;
        if (w < 2) {
            x = q28_sub_lazy<2>(m.t(2 * w + 1), m.t(2 * w));  // < 3.1 q
            y = x;
        } else {
            x = m.a(w);
            y = m.b(w);
        }
        const Fq28 r = q28_mul_lazy(x, y);
        m.set_t(4 + w, r);
        if (w < 2) m.set_t(2 * w + 1, x);
        if (w == 0) m.set_slow(q28_is_zero_mod_q(r));
    } else if (m.slow()) {
        return;
    } else if (stage == 2) {
        // PPP = P PP | Q = U1 PP | ZZ3 = ZZ1 ZZ2 PP
        if (w == 3) return;
        const Fq28 r = q28_mul_lazy(m.t(w == 0 ? 1 : (w == 1 ? 0 : 6)), m.t(4));
        if (w == 2) m.set_a(2, r);
        else m.set_t(w == 0 ? 1 : 0, r);
    } else if (stage == 3) {
        // X3 = RR - PPP - 2Q, R (Q - X3) | S1 PPP | ZZZ3 = ZZZ1 ZZZ2 PPP
        if (w == 3) return;
        Fq28 x, y;
        if (w == 0) {
            const Fq28 Q = m.t(0);
            const Fq28 X3 = q28_sub2_lazy<4>(m.t(5), m.t(1), Q);  // < 5.1 q
            m.set_a(0, X3);
            x = m.t(3);
            y = q28_sub_lazy<6>(Q, X3);                           // < 7.1 q
        } else {
            x = m.t(w == 1 ? 2 : 7);
            y = m.t(1);
        }
        const Fq28 r = q28_mul_lazy(x, y);
        if (w == 2) m.set_a(3, r);
        else m.set_t(w == 0 ? 4 : 6, r);
    } else {
        // Y3 = R (Q - X3) - S1 PPP   (< 3.1 q)
        if (w == 0) m.set_a(1, q28_sub_lazy<2>(m.t(4), m.t(6)));
    }
}

// a <- 2 a for a point that is not the identity (dbl-2008-s-1, a = 0).
//   temporaries: 0 V -> M (S - X3), 1 XX, 2 W, 3 S, 4 M^2, 5 W Y1
template <class M>
PLONK_HD void coop_dbl_stage(M &m, int stage, int w) {
    if (stage == 0) {
        // V = (2 Y1)^2 | XX = X1^2
        if (w >= 2) return;
        const Fq28 x = w == 0 ? q28_scale_lazy<2>(m.a(1)) : m.a(0);  // < 6.2 q
        m.set_t(w, q28_mul_lazy(x, x));
    } else if (stage == 1) {
        // W = U V | S = X1 V | M^2, M = 3 XX | ZZ3 = V ZZ1
        Fq28 x, y;
        if (w == 2) {
            x = q28_scale_lazy<3>(m.t(1));  // < 3.1 q
            y = x;
        } else {
            x = w == 0 ? q28_scale_lazy<2>(m.a(1)) : (w == 1 ? m.a(0) : m.a(2));
            y = m.t(0);
        }
        const Fq28 r = q28_mul_lazy(x, y);
        if (w == 3) m.set_a(2, r);
        else m.set_t(w == 0 ? 2 : (w == 1 ? 3 : 4), r);
    } else if (stage == 2) {
        // X3 = M^2 - 2 S, M (S - X3) | W Y1 | ZZZ3 = W ZZZ1
        if (w == 3) return;
        Fq28 x, y;
        if (w == 0) {
            const Fq28 S = m.t(3);
            const Fq28 X3 = q28_sub2_lazy<4>(m.t(4), q28_zero(), S);  // < 5.1 q
            m.set_a(0, X3);
            x = q28_scale_lazy<3>(m.t(1));
            y = q28_sub_lazy<6>(S, X3);
        } else {
            x = m.t(2);
            y = m.a(w == 1 ? 1 : 3);
        }
        const Fq28 r = q28_mul_lazy(x, y);
        if (w == 2) m.set_a(3, r);
        else m.set_t(w == 0 ? 0 : 5, r);
    } else {
        // Y3 = M (S - X3) - W Y1   (< 3.1 q)
        if (w == 0) m.set_a(1, q28_sub_lazy<2>(m.t(0), m.t(5)));
    }
}

}  // namespace plonk
