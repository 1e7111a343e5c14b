"""Both halves of the reference's poly-IOP gadgets (plonk/src/common/protocols.rs) on limb arrays, every
polynomial operation and every group operation done by the HIP library through `Backend`.

Same names and argument order as the reference; `gp: &GlobalParameters` becomes the `Backend` whose context
holds the SRS.  Polynomials and vectors are (n, 4) uint64 Montgomery limbs (`Vec<Fr>` as it lies in memory),
field elements (4,) limbs, group elements (18,) limbs.  Proofs are dictionaries with the field names of the
reference's proof structs (protocols.rs:17-123).

What the reference computes point by point and from scratch (the running products of
`compute_t_and_t1_*`: O(k^2) evaluations and inversions, protocols.rs:243-250, 345-353, 461-468, 571-588) runs
as one evaluation over Omega, one batched inversion and one scan; the results are the same field elements.
The extra interpolation points come from `ark_std::test_rng()` exactly where the reference draws them."""
import numpy as np

from .ark_rng import TestRng
from .prover import R_MOD, _m, from_mont, to_mont, trim


def _is_pow2(k):
    return k >= 1 and k & (k - 1) == 0


def degree(f):
    """`f.degree()` of ark's DensePolynomial: 0 for the zero polynomial."""
    return max(trim(np.asarray(f, dtype=np.uint64).reshape(-1, 4)).shape[0] - 1, 0)


def _evaluate_over(be, f, Omega):
    """f(Omega[j]) for every j (`f.evaluate(&Omega[j])`)."""
    k = Omega.shape[0]
    if f.shape[0] == 0:
        return np.zeros((k, 4), dtype=np.uint64)
    if _is_pow2(k) and k >= 2:
        return be.fr_evaluate_over_omega(f, k.bit_length() - 1)  # Omega = construct_Omega(k)
    return be.fr_eval_many(f, Omega)


def _interpolate_t(be, Omega, t_y, f):
    """t through (Omega[j], t_y[j]) and f.degree() + 1 - k extra random points (protocols.rs:251-255): the values
    first, then the abscissae, from a fresh `test_rng()`."""
    k = Omega.shape[0]
    extra = degree(f) + 1 - k
    if extra < 0:
        raise OverflowError("attempt to subtract with overflow (f.degree() + 1 - k)")
    if extra == 0 and _is_pow2(k):
        return trim(be.interpolate_over_omega(t_y))
    rng = TestRng()
    ys = np.concatenate([t_y, rng.fr_vec(extra)])
    xs = np.concatenate([Omega, rng.fr_vec(extra)])
    return trim(be.interpolate_polynomial(xs, ys))


def _shift(be, f, w):
    """`compose_polynomials(f, [0, w])` = f(wX); the reference's composition panics on fewer than two
    coefficients (polynomials.rs:84-112: index / empty reduce)."""
    if f.shape[0] < 2:
        raise RuntimeError("compose_polynomials panics in the reference for len(f) < 2")
    return trim(be.compose_with_scaling(f, w))


def _lin(be, x, y, a, b):
    return trim(be.fr_linear_combination(x, y, _m(a), _m(b), _m(0)))


def _mul(be, a, b):
    return trim(be.fr_poly_mul(a, b))


def compute_q_zero_test(be, k, f):
    """protocols.rs:157-166: quotient of f by X^k - 1, remainder dropped."""
    return trim(be.compute_q_zero_test(k, f))


def compute_q_zero_test_from_roots(be, roots, f):
    """protocols.rs:169-181."""
    return trim(be.compute_q_zero_test_from_roots(roots, f))


def compute_t_and_t1_product_check(be, Omega, f):
    """protocols.rs:235-275: t(Omega[i]) = prod_{j<=i} f(Omega[j]); t1 = t(wX) - t f(wX)."""
    w = Omega[1]
    t = _interpolate_t(be, Omega, be.fr_prefix_product(_evaluate_over(be, f, Omega)), f)
    t1 = _lin(be, _shift(be, t, w), _mul(be, t, _shift(be, f, w)), 1, R_MOD - 1)
    return t, t1


def compute_t_and_t1_product_check_rational_functions(be, Omega, f, g):
    """protocols.rs:336-385: t(Omega[i]) = prod_{j<=i} f / g; t1 = t(wX) g(wX) - t f(wX).  A zero of g on Omega
    raises ReferencePanic (`inverse().unwrap()`, :349)."""
    w = Omega[1]
    t_y = be.fr_ratio_prefix_product(_evaluate_over(be, f, Omega), _evaluate_over(be, g, Omega))
    t = _interpolate_t(be, Omega, t_y, f)
    t1 = _lin(be, _mul(be, _shift(be, t, w), _shift(be, g, w)), _mul(be, t, _shift(be, f, w)), 1, R_MOD - 1)
    return t, t1


def compute_t_and_t1_sum_check(be, Omega, f):
    """protocols.rs:453-493: t(Omega[i]) = sum_{j<=i} f(Omega[j]); t1 = t(wX) - (t + f(wX))."""
    w = Omega[1]
    t = _interpolate_t(be, Omega, be.fr_prefix_sum(_evaluate_over(be, f, Omega)), f)
    t1 = _lin(be, _shift(be, t, w), _lin(be, t, _shift(be, f, w), 1, 1), 1, R_MOD - 1)
    return t, t1


def compute_t_and_t1_prescribed_permutation_check(be, Omega, f, g, W, r, s):
    """protocols.rs:559-638: t(Omega[i]) = prod_{j<=i} (r - s W(w^j) - f(w^j)) / (r - s w^j - g(w^j));
    t1 = t(wX)(r - s w X - g(wX)) - t (r - s W(wX) - f(wX)).  A zero denominator on Omega raises ReferencePanic
    (`inverse().unwrap()`, :583)."""
    w = Omega[1]
    minus_s = _m((R_MOD - _i(s)) % R_MOD)
    minus_1 = _m(R_MOD - 1)
    num = be.fr_linear_combination(_evaluate_over(be, W, Omega), _evaluate_over(be, f, Omega), minus_s, minus_1, r,
                                   broadcast=True)
    den = be.fr_linear_combination(Omega, _evaluate_over(be, g, Omega), minus_s, minus_1, r, broadcast=True)
    t = _interpolate_t(be, Omega, be.fr_ratio_prefix_product(num, den), f)
    r_s_w_x = np.stack([np.asarray(r, dtype=np.uint64).reshape(4), _fr_mul(minus_s, w)])
    a = trim(be.fr_linear_combination(r_s_w_x, _shift(be, g, w), _m(1), minus_1, _m(0)))
    b = trim(be.fr_linear_combination(_shift(be, W, w), _shift(be, f, w), minus_s, minus_1, r))
    t1 = _lin(be, _mul(be, _shift(be, t, w), a), _mul(be, t, b), 1, R_MOD - 1)
    return t, t1


def compute_t1_T_S_zero_test(be, w, T, S):
    """protocols.rs:722-750: t1 = S (T + T(wY)) + (1 - S) T T(wY) - T(w^2 Y)."""
    T_w = _shift(be, T, w)
    T_w2 = _shift(be, T, _fr_mul(w, w))
    if S.shape[0] < 2:
        raise RuntimeError("compose_polynomials panics in the reference for len(S) < 2")
    one_minus_S = trim(be.fr_linear_combination(S, S[:0], _m(R_MOD - 1), _m(0), _m(1)))
    first = _mul(be, S, _lin(be, T, T_w, 1, 1))
    second = _mul(be, _mul(be, one_minus_S, T), T_w)
    return _lin(be, _lin(be, first, second, 1, 1), T_w2, 1, R_MOD - 1)


def _open(be, f, points):
    vs, proofs = be.kzg_evaluate_many(f, np.stack(points))
    return [(vs[i], proofs[i]) for i in range(len(points))]


def _fr_mul(a, b):
    return to_mont([from_mont(a)[0] * from_mont(b)[0] % R_MOD])[0]


def _fr_pow(a, e):
    return to_mont([pow(from_mont(a)[0], e, R_MOD)])[0]


def prove_equality(be, f, g, r):
    """protocols.rs:126-141."""
    (y_f, proof_f), = _open(be, f, [r])
    (y_g, proof_g), = _open(be, g, [r])
    return dict(y_f=y_f, proof_f=proof_f, y_g=y_g, proof_g=proof_g)


def prove_zero_test(be, f, q, r):
    """protocols.rs:184-204."""
    com_q = be.kzg_commit(q)
    (f_r, proof_f_r), = _open(be, f, [r])
    (q_r, proof_q_r), = _open(be, q, [r])
    return dict(com_q=com_q, f_r=f_r, proof_f_r=proof_f_r, q_r=q_r, proof_q_r=proof_q_r)


def _prove_t_q(be, w, k, t, q, r):
    # the commitments of one gadget in one batch, the openings of one polynomial in one call
    com_t, com_q = be.kzg_commit_batch([t, q])
    (t_last, p_last), (t_r, p_t_r), (t_w_r, p_t_w_r) = _open(be, t, [_fr_pow(w, k - 1), r, _fr_mul(r, w)])
    (q_r, p_q_r), = _open(be, q, [r])
    return dict(com_t=com_t, com_q=com_q, t_w_k_minus_1=t_last, proof_t_w_k_minus_1=p_last, t_r=t_r, proof_t_r=p_t_r,
                t_w_r=t_w_r, proof_t_w_r=p_t_w_r, q_r=q_r, proof_q_r=p_q_r)


def prove_product_check(be, w, k, t, q, f, r):
    """protocols.rs:278-309."""
    p = _prove_t_q(be, w, k, t, q, r)
    (p["f_w_r"], p["proof_f_w_r"]), = _open(be, f, [_fr_mul(r, w)])
    return p


def prove_sum_check(be, w, k, t, q, f, r):
    """protocols.rs:496-532."""
    return prove_product_check(be, w, k, t, q, f, r)


def prove_product_check_rational_functions(be, w, k, t, q, f, g, r):
    """protocols.rs:388-423."""
    p = prove_product_check(be, w, k, t, q, f, r)
    (p["g_w_r"], p["proof_g_w_r"]), = _open(be, g, [_fr_mul(r, w)])
    return p


def prove_prescribed_permutation_check(be, w, k, t, q, f, g, W, rp):
    """protocols.rs:641-687."""
    c = _prove_t_q(be, w, k, t, q, rp)
    wrp = _fr_mul(rp, w)
    p = dict(com_t=c["com_t"], com_q=c["com_q"], t_w_k_minus_1=c["t_w_k_minus_1"],
             proof_t_w_k_minus_1=c["proof_t_w_k_minus_1"], t_rp=c["t_r"], proof_t_rp=c["proof_t_r"],
             t_w_rp=c["t_w_r"], proof_t_w_rp=c["proof_t_w_r"], q_rp=c["q_r"], proof_q_rp=c["proof_q_r"])
    for name, poly in (("f", f), ("g", g), ("W", W)):
        (p[name + "_w_rp"], p["proof_" + name + "_w_rp"]), = _open(be, poly, [wrp])
    return p


def prove_T_S_zero_test(be, w, q, T, S, r):
    """protocols.rs:753-786."""
    com_q = be.kzg_commit(q)
    wr = _fr_mul(w, r)
    (T_r, p_T_r), (T_w_r, p_T_w_r), (T_w2_r, p_T_w2_r) = _open(be, T, [r, wr, _fr_mul(w, wr)])
    (S_r, p_S_r), = _open(be, S, [r])
    (q_r, p_q_r), = _open(be, q, [r])
    return dict(com_q=com_q, T_r=T_r, proof_T_r=p_T_r, T_w_r=T_w_r, proof_T_w_r=p_T_w_r, T_w2_r=T_w2_r,
                proof_T_w2_r=p_T_w2_r, S_r=S_r, proof_S_r=p_S_r, q_r=q_r, proof_q_r=p_q_r)


# ----------------------------------------------------------------------------------------------------------
# Verifier halves: the field equation of each gadget and its `kzg_verify` calls (kzg.rs:91-98) folded into ONE
# pairing-product check (host code of the library).
# ----------------------------------------------------------------------------------------------------------
def _pairings(tau_g2, checks):
    from . import backend as _b
    coms, us, vs, proofs = zip(*checks)
    return _b.kzg_verify_batch(tau_g2, np.stack(coms), np.stack(us), np.stack(vs), np.stack(proofs))


def _i(x):
    return from_mont(x)[0]


def verify_equality(tau_g2, com_f, com_g, r, p):
    """protocols.rs:144-154."""
    return _i(p["y_f"]) == _i(p["y_g"]) and _pairings(tau_g2, [(com_f, r, p["y_f"], p["proof_f"]),
                                                               (com_g, r, p["y_g"], p["proof_g"])])


def verify_zero_test(tau_g2, k, com_f, r, p):
    """protocols.rs:207-217."""
    z = (pow(_i(r), k, R_MOD) - 1) % R_MOD
    return _i(p["f_r"]) == _i(p["q_r"]) * z % R_MOD and _pairings(
        tau_g2, [(p["com_q"], r, p["q_r"], p["proof_q_r"]), (com_f, r, p["f_r"], p["proof_f_r"])])


def verify_zero_on_roots_test(be, tau_g2, roots, com_f, r, p):
    """protocols.rs:220-232."""
    z = _i(be.fr_eval(be.construct_vanishing_polynomial_from_roots(roots), r))
    return _i(p["f_r"]) == _i(p["q_r"]) * z % R_MOD and _pairings(
        tau_g2, [(p["com_q"], r, p["q_r"], p["proof_q_r"]), (com_f, r, p["f_r"], p["proof_f_r"])])


def _verify_t_q(tau_g2, w, k, r, p, last, lhs, more):
    z = (pow(_i(r), k, R_MOD) - 1) % R_MOD
    if _i(p["t_w_k_minus_1"]) != last or lhs % R_MOD != _i(p["q_r"]) * z % R_MOD:
        return False
    rw = _fr_mul(r, w)
    return _pairings(tau_g2, [(p["com_t"], _fr_pow(w, k - 1), p["t_w_k_minus_1"], p["proof_t_w_k_minus_1"]),
                              (p["com_t"], r, p["t_r"], p["proof_t_r"]), (p["com_t"], rw, p["t_w_r"], p["proof_t_w_r"]),
                              (p["com_q"], r, p["q_r"], p["proof_q_r"])] + [(c, rw, p[v], p[pr]) for c, v, pr in more])


def verify_product_check(tau_g2, w, k, com_f, r, p):
    """protocols.rs:312-333."""
    return _verify_t_q(tau_g2, w, k, r, p, 1, _i(p["t_w_r"]) - _i(p["t_r"]) * _i(p["f_w_r"]),
                       [(com_f, "f_w_r", "proof_f_w_r")])


def verify_product_check_rational_functions(tau_g2, w, k, com_f, com_g, r, p):
    """protocols.rs:426-450."""
    return _verify_t_q(tau_g2, w, k, r, p, 1, _i(p["t_w_r"]) * _i(p["g_w_r"]) - _i(p["t_r"]) * _i(p["f_w_r"]),
                       [(com_f, "f_w_r", "proof_f_w_r"), (com_g, "g_w_r", "proof_g_w_r")])


def verify_sum_check(tau_g2, w, k, com_f, r, p):
    """protocols.rs:535-556."""
    return _verify_t_q(tau_g2, w, k, r, p, 0, _i(p["t_w_r"]) - (_i(p["t_r"]) + _i(p["f_w_r"])),
                       [(com_f, "f_w_r", "proof_f_w_r")])


def verify_prescribed_permutation_check(tau_g2, w, k, com_f, com_g, com_W, r, s, rp, p):
    """protocols.rs:690-719."""
    z = (pow(_i(rp), k, R_MOD) - 1) % R_MOD
    lhs = (_i(p["t_w_rp"]) * (_i(r) - _i(s) * _i(w) * _i(rp) - _i(p["g_w_rp"]))
           - _i(p["t_rp"]) * (_i(r) - _i(s) * _i(p["W_w_rp"]) - _i(p["f_w_rp"]))) % R_MOD
    if _i(p["t_w_k_minus_1"]) != 1 or lhs != _i(p["q_rp"]) * z % R_MOD:
        return False
    wrp = _fr_mul(rp, w)
    return _pairings(tau_g2, [(p["com_t"], _fr_pow(w, k - 1), p["t_w_k_minus_1"], p["proof_t_w_k_minus_1"]),
                              (p["com_t"], rp, p["t_rp"], p["proof_t_rp"]),
                              (p["com_t"], wrp, p["t_w_rp"], p["proof_t_w_rp"]),
                              (p["com_q"], rp, p["q_rp"], p["proof_q_rp"]),
                              (com_f, wrp, p["f_w_rp"], p["proof_f_w_rp"]),
                              (com_g, wrp, p["g_w_rp"], p["proof_g_w_rp"]),
                              (com_W, wrp, p["W_w_rp"], p["proof_W_w_rp"])])


def verify_T_S_zero_test(be, tau_g2, w, Omega_gates, com_T, com_S, r, p):
    """protocols.rs:789-809."""
    z = _i(be.fr_eval(be.construct_vanishing_polynomial_from_roots(Omega_gates), r))
    S_r, T_r, T_w_r = _i(p["S_r"]), _i(p["T_r"]), _i(p["T_w_r"])
    lhs = (S_r * (T_r + T_w_r) + (1 - S_r) * T_r * T_w_r - _i(p["T_w2_r"])) % R_MOD
    if lhs != _i(p["q_r"]) * z % R_MOD:
        return False
    wr = _fr_mul(w, r)
    return _pairings(tau_g2, [(com_T, r, p["T_r"], p["proof_T_r"]), (com_T, wr, p["T_w_r"], p["proof_T_w_r"]),
                              (com_T, _fr_mul(w, wr), p["T_w2_r"], p["proof_T_w2_r"]),
                              (com_S, r, p["S_r"], p["proof_S_r"]), (p["com_q"], r, p["q_r"], p["proof_q_r"])])
