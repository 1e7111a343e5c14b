// A compiled host program on plonk_hip.hpp: the shape of plonk/tests/kzg_tests.rs (setup, commit, open, verify,
// SRS too short) plus an interpolation round trip and the polynomial helpers under the reference's names.  Input on the command line: the trapdoor and the point
// as 4 limbs each (Montgomery form, hex), then the coefficients as raw limbs on stdin.  Prints the
// compressed commitment, the opened value (limbs) and the compressed proof as hex lines; the test harness
// compares them with the oracle.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "plonk_hip.hpp"

using namespace plonk_hip;

static Fr parse_fr(char **argv) {
    Fr x;
    for (int i = 0; i < 4; i++) x[i] = std::strtoull(argv[i], nullptr, 16);
    return x;
}
static void print_hex(const uint8_t *p, size_t n) {
    for (size_t i = 0; i < n; i++) std::printf("%02x", p[i]);
    std::printf("\n");
}

int main(int argc, char **argv) {
    if (argc != 10) {
        std::fprintf(stderr, "usage: kzg_roundtrip DEGREE tau0 tau1 tau2 tau3 u0 u1 u2 u3 < coefficients\n");
        return 2;
    }
    const size_t degree = std::strtoull(argv[1], nullptr, 10);
    const Fr tau = parse_fr(argv + 2), u = parse_fr(argv + 6);
    DensePolynomial f;
    Fr c;
    while (std::fread(c.data(), 8, 4, stdin) == 4) f.push_back(c);
    try {
        GlobalParameters gp = kzg_setup(degree, tau);
        std::printf("srs %zu\n", gp.size());
        G1 com = kzg_commit(gp, f);
        auto [v, proof] = kzg_evaluate(gp, f, u);
        auto enc = g1_compress(gp, {com, proof});
        print_hex(enc[0].data(), 48);
        std::printf("%016llx %016llx %016llx %016llx\n", (unsigned long long)v[0], (unsigned long long)v[1],
                    (unsigned long long)v[2], (unsigned long long)v[3]);
        print_hex(enc[1].data(), 48);
        // plonk/tests/kzg_tests.rs:148-197: the opening verifies by pairings against gp.tau_g2, a wrong value does not
        const G2Compressed tau_g2 = kzg_setup_tau_g2(tau);
        Fr wrong = v;
        wrong[0] ^= 1;
        const bool verified = kzg_verify(tau_g2, com, u, v, proof) && !kzg_verify(tau_g2, com, u, wrong, proof);
        // interpolation over Omega is the inverse of evaluation over Omega
        size_t n = 1;
        while (n < f.size()) n *= 2;
        DensePolynomial back = interpolate_over_omega(gp, evaluate_over_omega(gp, f, n));
        back.resize(f.size());
        std::printf("roundtrip %s\n", back == f ? "ok" : "MISMATCH");
        std::printf("verify %s\n", verified ? "ok" : "FAILED");
        // the polynomial layer under the reference's names: Z = prod (X - r_i) over the first coefficients taken
        // as roots vanishes on them, (f * Z) / Z gives f back (protocols.rs:169-181), the interpolation through
        // (r_i, f(r_i)) is f again when f is that short (polynomials.rs:14-58), and f(wX) at u equals f at w u
        // (polynomials.rs:80-113)
        const size_t m = f.size() < 24 ? f.size() : 24;
        const std::vector<Fr> roots(f.begin(), f.begin() + m);
        const DensePolynomial z = construct_vanishing_polynomial_from_roots(gp, roots);
        bool poly_ok = z.size() == m + 1;
        for (const Fr &r : roots) poly_ok = poly_ok && evaluate(gp, z, r) == Fr{0, 0, 0, 0};
        DensePolynomial q = compute_q_zero_test_from_roots(gp, roots, multiply(gp, f, z));
        poly_ok = poly_ok && q == f;
        const DensePolynomial low(f.begin(), f.begin() + m);
        std::vector<Fr> ys;
        for (const Fr &r : roots) ys.push_back(evaluate(gp, low, r));
        poly_ok = poly_ok && interpolate_polynomial(gp, roots, ys) == low;
        const Fr wu = multiply(gp, {tau}, {u})[0];  // tau * u as a product of constants
        poly_ok = poly_ok && evaluate(gp, compose_with_scaled_argument(gp, f, tau), u) == evaluate(gp, f, wu);
        std::printf("polynomials %s\n", poly_ok ? "ok" : "MISMATCH");
        // plonk/tests/kzg_tests.rs:71-93: a polynomial longer than the SRS is a CommitError
        DensePolynomial too_long(degree + 2, Fr{1, 0, 0, 0});
        try {
            kzg_commit(gp, too_long);
            std::printf("commit_error MISSING\n");
        } catch (const CommitError &e) {
            std::printf("commit_error %s\n", e.what());
        }
        try {
            kzg_commit(gp, DensePolynomial{});
            std::printf("panic MISSING\n");
        } catch (const ReferencePanic &) {
            std::printf("panic ok\n");
        }
    } catch (const Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
