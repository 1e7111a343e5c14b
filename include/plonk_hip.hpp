// C++ host side over the C ABI of plonk_hip.h: the functions of the reference's `common::kzg` and
// `common::polynomials` with their names, argument meaning and error behaviour, for callers written
// in a compiled language (the reference is Rust; a Rust shim binds the same C entry points).
//
//   kzg_setup(degree)            plonk/src/common/kzg.rs:24-45      -> GlobalParameters (SRS on the device)
//   kzg_commit(gp, f)            kzg.rs:48-60                       -> G1, or CommitError
//   kzg_evaluate(gp, f, u)       kzg.rs:63-88                       -> (Fr, G1)
//   interpolate_over_omega(ys)   plonk/src/common/polynomials.rs:50-58 with x = construct_Omega(n)
//   compute_q_zero_test(k, f)    plonk/src/common/protocols.rs:157-166
//   interpolate_polynomial(xs, ys)                      polynomials.rs:14-58, any points
//   compose_with_scaled_argument(f, w)                  polynomials.rs:80-113 as the gadgets call it: f(wX)
//   construct_vanishing_polynomial_from_roots(roots)    plonk/src/common/utils.rs:51-62
//   compute_q_zero_test_from_roots(roots, f)            protocols.rs:169-181
//   multiply(a, b), evaluate(f, u)                      `&a * &b`, `f.evaluate(&u)`
//
// Fr is ark-ff's in-memory form (4 x u64 Montgomery limbs), G1 ark-ec's G1Projective (18 x u64).
// Where the reference returns Err the wrapper throws CommitError; where the reference panics it throws
// ReferencePanic; anything else (HIP failures, bad arguments) is a plain Error.  Header only.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "plonk_hip.h"

namespace plonk_hip {

using Fr = std::array<uint64_t, 4>;
using G1 = std::array<uint64_t, 18>;
using DensePolynomial = std::vector<Fr>;  // coefficients, lowest degree first

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};
// `CommitError::CommitFailed` (kzg.rs:11-15)
struct CommitError : Error {
    using Error::Error;
};
// inputs on which the reference panics (unwrap of None, index out of bounds, inverse of zero)
struct ReferencePanic : Error {
    using Error::Error;
};

// `GlobalParameters` (kzg.rs:17-21): owns the context that keeps tau_powers_g1 resident on one GPU.
class GlobalParameters {
  public:
    explicit GlobalParameters(int device = 0) {
        int rc = plonk_hip_ctx_create(device, &ctx_);
        if (rc != PLONK_HIP_OK) throw Error(rc, plonk_hip_strerror(rc));
    }
    ~GlobalParameters() {
        if (ctx_) plonk_hip_ctx_destroy(ctx_);
    }
    GlobalParameters(const GlobalParameters &) = delete;
    GlobalParameters &operator=(const GlobalParameters &) = delete;
    GlobalParameters(GlobalParameters &&o) noexcept : ctx_(std::exchange(o.ctx_, nullptr)) {}

    plonk_hip_ctx *ctx() const { return ctx_; }

    // tau_powers_g1 from a caller that already has them (n x 18 limbs)
    void load(const std::vector<G1> &tau_powers_g1) {
        check(plonk_hip_srs_load_projective(ctx_, tau_powers_g1.empty() ? nullptr : tau_powers_g1[0].data(),
                                            tau_powers_g1.size()));
    }
    size_t size() const {
        size_t n = 0;
        check(plonk_hip_srs_size(ctx_, &n));
        return n;
    }
    std::vector<G1> tau_powers_g1(size_t first, size_t count) const {
        std::vector<G1> out(count);
        if (count) check(plonk_hip_srs_read(ctx_, first, count, out[0].data()));
        return out;
    }

    void check(int rc, bool commit = false) const {
        if (rc == PLONK_HIP_OK) return;
        std::string text = plonk_hip_strerror(rc);
        if (rc == PLONK_HIP_ERR_HIP) text += std::string(": ") + plonk_hip_last_error(ctx_);
        if (rc == PLONK_HIP_ERR_SRS_TOO_SHORT && commit) throw CommitError(rc, text);
        if (rc == PLONK_HIP_ERR_SRS_TOO_SHORT || rc == PLONK_HIP_ERR_EMPTY_POLY || rc == PLONK_HIP_ERR_ZERO_DENOMINATOR ||
            rc == PLONK_HIP_ERR_INVALID_POINT)
            throw ReferencePanic(rc, text);
        throw Error(rc, text);
    }

  private:
    plonk_hip_ctx *ctx_ = nullptr;
};

// `kzg_setup(degree)` with the trapdoor given by the caller (the reference draws it from
// ark_std::test_rng(), kzg.rs:25-28): [tau^i]G for i = 0..degree, resident on `device`.
inline GlobalParameters kzg_setup(size_t degree, const Fr &tau, int device = 0) {
    GlobalParameters gp(device);
    gp.check(plonk_hip_kzg_setup(gp.ctx(), tau.data(), degree, nullptr));
    return gp;
}

inline G1 kzg_commit(const GlobalParameters &gp, const DensePolynomial &f) {
    G1 out;
    gp.check(plonk_hip_kzg_commit(gp.ctx(), f.empty() ? nullptr : f[0].data(), f.size(), out.data()), true);
    return out;
}

inline std::pair<Fr, G1> kzg_evaluate(const GlobalParameters &gp, const DensePolynomial &f, const Fr &u) {
    Fr v;
    G1 proof;
    gp.check(plonk_hip_kzg_open(gp.ctx(), f.empty() ? nullptr : f[0].data(), f.size(), u.data(), v.data(), proof.data()));
    return {v, proof};
}

// gp.tau_g2 (kzg.rs:20, 38-39) in its 96-byte compressed form, and kzg_verify (kzg.rs:91-98) on it
using G2Compressed = std::array<uint8_t, 96>;
inline G2Compressed kzg_setup_tau_g2(const Fr &tau) {
    G2Compressed out;
    if (int rc = plonk_hip_kzg_setup_tau_g2(tau.data(), out.data())) throw Error(rc, plonk_hip_strerror(rc));
    return out;
}
inline bool kzg_verify(const G2Compressed &tau_g2, const G1 &com_f, const Fr &u, const Fr &v, const G1 &proof) {
    int ok = 0;
    if (int rc = plonk_hip_kzg_verify(tau_g2.data(), com_f.data(), u.data(), v.data(), proof.data(), &ok))
        throw Error(rc, plonk_hip_strerror(rc));
    return ok != 0;
}
// every (com_f, u, v, proof) at once; rho: one 128-bit weight per check (two words each), drawn by the caller
// after the proofs are fixed
inline bool kzg_verify_batch(const G2Compressed &tau_g2, const std::vector<G1> &coms, const std::vector<Fr> &us,
                             const std::vector<Fr> &vs, const std::vector<G1> &proofs,
                             const std::vector<std::array<uint64_t, 2>> &rho) {
    const size_t n = coms.size();
    if (us.size() != n || vs.size() != n || proofs.size() != n || rho.size() < n)
        throw Error(PLONK_HIP_ERR_INVALID_ARG, plonk_hip_strerror(PLONK_HIP_ERR_INVALID_ARG));
    int ok = 0;
    if (int rc = plonk_hip_kzg_verify_batch(tau_g2.data(), n, n ? coms[0].data() : nullptr, n ? us[0].data() : nullptr,
                                            n ? vs[0].data() : nullptr, n ? proofs[0].data() : nullptr,
                                            n ? rho[0].data() : nullptr, &ok))
        throw Error(rc, plonk_hip_strerror(rc));
    return ok != 0;
}

// interpolate_polynomial(construct_Omega(n), ys) for n = 2^k or 3 * 2^k
inline DensePolynomial interpolate_over_omega(const GlobalParameters &gp, std::vector<Fr> ys) {
    if (!ys.empty()) gp.check(plonk_hip_fr_ntt_mixed(gp.ctx(), ys[0].data(), ys.size(), 1, 1));
    return ys;
}

// [f(Omega[j]) for j < n], f of at most n coefficients
inline std::vector<Fr> evaluate_over_omega(const GlobalParameters &gp, DensePolynomial f, size_t n) {
    f.resize(n, Fr{0, 0, 0, 0});
    if (n) gp.check(plonk_hip_fr_ntt_mixed(gp.ctx(), f[0].data(), n, 1, 0));
    return f;
}

inline DensePolynomial compute_q_zero_test(const GlobalParameters &gp, size_t k, const DensePolynomial &f) {
    DensePolynomial q(f.size() > k ? f.size() - k : 0);
    if (!q.empty()) gp.check(plonk_hip_fr_div_vanishing(gp.ctx(), f[0].data(), f.size(), k, q[0].data()));
    return q;
}

// interpolate_polynomial(x_points, y_points) (polynomials.rs:14-58): the polynomial of degree below n through n
// points; two equal x are a ReferencePanic (the reference's `.inverse().unwrap()`, polynomials.rs:42)
inline DensePolynomial interpolate_polynomial(const GlobalParameters &gp, const std::vector<Fr> &xs,
                                              const std::vector<Fr> &ys) {
    if (xs.size() != ys.size()) throw Error(PLONK_HIP_ERR_INVALID_ARG, plonk_hip_strerror(PLONK_HIP_ERR_INVALID_ARG));
    DensePolynomial out(xs.size());
    if (!xs.empty()) gp.check(plonk_hip_fr_interpolate(gp.ctx(), xs[0].data(), ys[0].data(), xs.size(), out[0].data()));
    return out;
}

// compose_polynomials(f, [0, w]) (polynomials.rs:80-113; the only inner polynomial the gadgets use,
// protocols.rs:258-270 and the like): f(wX)
inline DensePolynomial compose_with_scaled_argument(const GlobalParameters &gp, DensePolynomial f, const Fr &w) {
    if (!f.empty()) gp.check(plonk_hip_fr_scale_by_powers(gp.ctx(), f[0].data(), f.size(), w.data()));
    return f;
}

// f.evaluate(&u)
inline Fr evaluate(const GlobalParameters &gp, const DensePolynomial &f, const Fr &u) {
    Fr out{0, 0, 0, 0};
    if (!f.empty()) gp.check(plonk_hip_fr_eval(gp.ctx(), f[0].data(), f.size(), u.data(), out.data()));
    return out;
}

// &a * &b (untrimmed: a.size() + b.size() - 1 coefficients)
inline DensePolynomial multiply(const GlobalParameters &gp, const DensePolynomial &a, const DensePolynomial &b) {
    if (a.empty() || b.empty()) return {};
    DensePolynomial out(a.size() + b.size() - 1);
    gp.check(plonk_hip_fr_poly_mul(gp.ctx(), a[0].data(), a.size(), b[0].data(), b.size(), out[0].data()));
    return out;
}

// construct_vanishing_polynomial_from_roots(roots) (utils.rs:51-62): prod (X - roots[i])
inline DensePolynomial construct_vanishing_polynomial_from_roots(const GlobalParameters &gp, const std::vector<Fr> &roots) {
    DensePolynomial z(roots.size() + 1);
    gp.check(plonk_hip_fr_vanishing_from_roots(gp.ctx(), roots.empty() ? nullptr : roots[0].data(), roots.size(),
                                               z[0].data()));
    return z;
}

// compute_q_zero_test_from_roots(roots, f) (protocols.rs:169-181): f / prod (X - roots[i]), remainder dropped
inline DensePolynomial compute_q_zero_test_from_roots(const GlobalParameters &gp, const std::vector<Fr> &roots,
                                                      const DensePolynomial &f) {
    DensePolynomial q(f.size() > roots.size() ? f.size() - roots.size() : 0);
    if (!q.empty())
        gp.check(plonk_hip_fr_div_by_roots(gp.ctx(), f[0].data(), f.size(), roots.empty() ? nullptr : roots[0].data(),
                                           roots.size(), q[0].data()));
    return q;
}

// compressed encodings (`serialize_compressed`, plonk/src/common/utils.rs:70-72)
inline std::vector<std::array<uint8_t, 48>> g1_compress(const GlobalParameters &gp, const std::vector<G1> &pts) {
    std::vector<std::array<uint8_t, 48>> out(pts.size());
    if (!pts.empty()) gp.check(plonk_hip_g1_compress(gp.ctx(), pts[0].data(), pts.size(), out[0].data()));
    return out;
}

}  // namespace plonk_hip
