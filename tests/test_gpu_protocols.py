"""GPU parity of the gadgets as the reference's own tests drive them -- equality, zero test, zero on roots, product
check, sum check, product check of rational functions, permutation check, prescribed permutation check, T_S zero
test (plonk/src/common/protocols.rs:126-809) -- end to end through the C ABI (plonk_amd/protocols.py) against the
oracle's restatement, on the reference's own test inputs regenerated draw by draw
(plonk/tests/protocols_tests.rs:24-1149: k = 8, deg f = 10, so compute_t_and_t1_* draw three extra interpolation
points from ark_std::test_rng(); the permutations come from the restated, unpinned shuffle of oracle/arkrng.py).  Bit-exact: every polynomial coefficient, opened
value and compressed group element; the verdicts are the reference's (success cases verify, fail cases do not),
by the library's pairing check and by the oracle's."""
import numpy as np
import pytest

import plonk_amd
from oracle import arkrng as a
from oracle import capi as c
from oracle import pyref as o
from plonk_amd import protocols as P
from tests import protocol_flows as flows
from tests.gpu_util import backend, compress  # noqa: F401

pytestmark = pytest.mark.gpu


def limbs(values):
    return c.fr_to_mont(list(values))


def ints(arr):
    return c.fr_from_mont(np.ascontiguousarray(arr).reshape(-1, 4))


def same_proof(got, want):
    assert set(got) == set(want)
    for name, w in want.items():
        g = got[name]
        if isinstance(w, int):
            assert ints(g)[0] == w, name
        else:
            assert compress(g).hex() == o.g1_hex(w), name


@pytest.fixture(scope="module")
def srs_of_the_reference(backend, golden):
    def setup(degree):
        pts, tau_g2 = backend.kzg_setup_own_tau(degree)
        gp, tau = a.kzg_setup(degree)
        assert [bytes(e).hex() for e in c.g1_compress(pts)] == [o.g1_hex(p) for p in gp.tau_powers_g1]
        return gp, tau, tau_g2
    return setup


def test_kzg_setup_with_its_own_tau_is_the_golden_srs(backend, golden):
    # setup_global_params::run -> kzg_setup(21) (setup_global_params/mod.rs:60-61) = plonk/data/srs.json
    pts, tau_g2 = backend.kzg_setup_own_tau(21)
    assert [bytes(e).hex() for e in c.g1_compress(pts)] == golden["srs"]["gp"]["tau_powers_g1"]
    assert bytes(tau_g2).hex() == golden["srs"]["gp"]["tau_g2"]


@pytest.mark.parametrize("success", [True, False])
@pytest.mark.parametrize("gadget", ["product_check", "sum_check", "rational_check"])
def test_running_product_gadgets_end_to_end(backend, srs_of_the_reference, gadget, success):
    case = getattr(flows, gadget)(success)
    gp, tau, tau_g2 = srs_of_the_reference(case["degree"])
    Omega, f, r, k = case["Omega"], case["f"], case["r"], flows.K
    w = Omega[1]
    mOmega, mf, mr, mw = limbs(Omega), limbs(f), limbs([r])[0], limbs([w])[0]
    com_f = backend.kzg_commit(mf)
    assert compress(com_f).hex() == o.g1_hex(o.kzg_commit(gp, f))
    if gadget == "product_check":
        t, t1 = P.compute_t_and_t1_product_check(backend, mOmega, mf)
        want_t, want_t1 = o.compute_t_and_t1_product_check(Omega, f)
    elif gadget == "sum_check":
        t, t1 = P.compute_t_and_t1_sum_check(backend, mOmega, mf)
        want_t, want_t1 = o.compute_t_and_t1_sum_check(Omega, f)
    else:
        mg = limbs(case["g"])
        com_g = backend.kzg_commit(mg)
        t, t1 = P.compute_t_and_t1_product_check_rational_functions(backend, mOmega, mf, mg)
        want_t, want_t1 = o.compute_t_and_t1_product_check_rational_functions(Omega, f, case["g"])
    assert ints(t) == o.poly_trim(want_t) and ints(t1) == o.poly_trim(want_t1)
    assert P.degree(t) == 10
    q = P.compute_q_zero_test(backend, k, t1)
    want_q = o.compute_q_zero_test(k, want_t1)
    assert ints(q) == o.poly_trim(want_q)
    # `assert_eq!(&q * &Z_Omega, t1)` / `assert_ne!`: q Z through the library's product
    z = limbs(o.construct_vanishing_polynomial(k))
    qz = ints(P.trim(backend.fr_poly_mul(q, z))) if q.shape[0] else []
    assert (qz == ints(t1)) == success
    check = o.kzg_verify_with_tau(tau)
    if gadget == "product_check":
        proof = P.prove_product_check(backend, mw, k, t, q, mf, mr)
        want = o.prove_product_check(gp, w, k, want_t, want_q, f, r)
        verdict = P.verify_product_check(tau_g2, mw, k, com_f, mr, proof)
        assert o.verify_product_check(check, w, k, o.kzg_commit(gp, f), r, want) == success
    elif gadget == "sum_check":
        proof = P.prove_sum_check(backend, mw, k, t, q, mf, mr)
        want = o.prove_sum_check(gp, w, k, want_t, want_q, f, r)
        verdict = P.verify_sum_check(tau_g2, mw, k, com_f, mr, proof)
        assert o.verify_sum_check(check, w, k, o.kzg_commit(gp, f), r, want) == success
    else:
        proof = P.prove_product_check_rational_functions(backend, mw, k, t, q, mf, mg, mr)
        want = o.prove_product_check_rational_functions(gp, w, k, want_t, want_q, f, case["g"], r)
        verdict = P.verify_product_check_rational_functions(tau_g2, mw, k, com_f, com_g, mr, proof)
        assert o.verify_product_check_rational_functions(check, w, k, o.kzg_commit(gp, f),
                                                         o.kzg_commit(gp, case["g"]), r, want) == success
    same_proof(proof, want)
    assert verdict == success


@pytest.mark.parametrize("success", [True, False])
def test_equality_zero_test_and_zero_on_roots_end_to_end(backend, srs_of_the_reference, success):
    gp, tau, tau_g2 = srs_of_the_reference(10)
    # protocols_tests.rs:24-80
    case = flows.equality(success)
    mf, mg, mr = limbs(case["f"]), limbs(case["g"]), limbs([case["r"]])[0]
    proof = P.prove_equality(backend, mf, mg, mr)
    same_proof(proof, o.prove_equality(gp, case["f"], case["g"], case["r"]))
    assert P.verify_equality(tau_g2, backend.kzg_commit(mf), backend.kzg_commit(mg), mr, proof) == success
    # :82-167
    case = flows.zero_test(success)
    mf, mr = limbs(case["f"]), limbs([case["r"]])[0]
    q = P.compute_q_zero_test(backend, flows.K, mf)
    want_q = o.compute_q_zero_test(flows.K, case["f"])
    assert ints(q) == o.poly_trim(want_q)
    proof = P.prove_zero_test(backend, mf, q, mr)
    same_proof(proof, o.prove_zero_test(gp, case["f"], want_q, case["r"]))
    assert P.verify_zero_test(tau_g2, flows.K, backend.kzg_commit(mf), mr, proof) == success
    # :169-251
    case = flows.zero_on_roots(success)
    mf, mr, mroots = limbs(case["f"]), limbs([case["r"]])[0], limbs(case["roots"])
    q = P.compute_q_zero_test_from_roots(backend, mroots, mf)
    want_q = o.compute_q_zero_test_from_roots(case["roots"], case["f"])
    assert ints(q) == o.poly_trim(want_q)
    proof = P.prove_zero_test(backend, mf, q, mr)
    same_proof(proof, o.prove_zero_test(gp, case["f"], want_q, case["r"]))
    assert P.verify_zero_on_roots_test(backend, tau_g2, mroots, backend.kzg_commit(mf), mr, proof) == success


@pytest.mark.parametrize("success", [True, False])
def test_permutation_and_prescribed_permutation_checks_end_to_end(backend, srs_of_the_reference, success):
    k = flows.K
    z = limbs(o.construct_vanishing_polynomial(k))
    # protocols_tests.rs:646-776: the permutation check is the product check of f / g
    case = flows.permutation_check(success)
    gp, tau, tau_g2 = srs_of_the_reference(case["degree"])
    check = o.kzg_verify_with_tau(tau)
    Omega, f, g, r, w = case["Omega"], case["f"], case["g"], case["r"], case["Omega"][1]
    mOmega, mf, mg, mr, mw = limbs(Omega), limbs(f), limbs(g), limbs([r])[0], limbs([w])[0]
    com_f, com_g = backend.kzg_commit_batch([mf, mg])
    t, t1 = P.compute_t_and_t1_product_check_rational_functions(backend, mOmega, mf, mg)
    want_t, want_t1 = o.compute_t_and_t1_product_check_rational_functions(Omega, f, g)
    assert ints(t) == o.poly_trim(want_t) and ints(t1) == o.poly_trim(want_t1)
    q = P.compute_q_zero_test(backend, k, t1)
    want_q = o.compute_q_zero_test(k, want_t1)
    assert ints(q) == o.poly_trim(want_q)
    assert (ints(P.trim(backend.fr_poly_mul(q, z))) == ints(t1)) == success
    proof = P.prove_product_check_rational_functions(backend, mw, k, t, q, mf, mg, mr)
    same_proof(proof, o.prove_product_check_rational_functions(gp, w, k, want_t, want_q, f, g, r))
    assert P.verify_product_check_rational_functions(tau_g2, mw, k, com_f, com_g, mr, proof) == success
    # :778-943
    case = flows.prescribed_permutation_check(success)
    f, g, W, r, s_, rp = (case[n] for n in ("f", "g", "W", "r", "s", "rp"))
    mf, mg, mW = limbs(f), limbs(g), limbs(W)
    mr, ms, mrp = limbs([r])[0], limbs([s_])[0], limbs([rp])[0]
    com_f, com_g, com_W = backend.kzg_commit_batch([mf, mg, mW])
    t, t1 = P.compute_t_and_t1_prescribed_permutation_check(backend, mOmega, mf, mg, mW, mr, ms)
    want_t, want_t1 = o.compute_t_and_t1_prescribed_permutation_check(Omega, f, g, W, r, s_)
    assert ints(t) == o.poly_trim(want_t) and ints(t1) == o.poly_trim(want_t1)
    q = P.compute_q_zero_test(backend, k, t1)
    want_q = o.compute_q_zero_test(k, want_t1)
    assert ints(q) == o.poly_trim(want_q)
    assert (ints(P.trim(backend.fr_poly_mul(q, z))) == ints(t1)) == success
    proof = P.prove_prescribed_permutation_check(backend, mw, k, t, q, mf, mg, mW, mrp)
    want = o.prove_prescribed_permutation_check(gp, w, k, want_t, want_q, f, g, W, rp)
    same_proof(proof, want)
    assert P.verify_prescribed_permutation_check(tau_g2, mw, k, com_f, com_g, com_W, mr, ms, mrp, proof) == success
    assert o.verify_prescribed_permutation_check(check, w, k, o.kzg_commit(gp, f), o.kzg_commit(gp, g),
                                                 o.kzg_commit(gp, W), r, s_, rp, want) == success
    if success:  # a tampered opening is caught by the pairings, a tampered value by the field equation
        bad = dict(proof)
        bad["W_w_rp"] = limbs([(ints(proof["W_w_rp"])[0] + 1) % o.R])[0]
        assert not P.verify_prescribed_permutation_check(tau_g2, mw, k, com_f, com_g, com_W, mr, ms, mrp, bad)


@pytest.mark.parametrize("success", [True, False])
def test_T_S_zero_test_end_to_end(backend, srs_of_the_reference, success):
    # protocols_tests.rs:945-1149: d = 12, three gates, degrees 24 and 21 as the reference asserts (:1037, :1058)
    case = flows.t_s_zero_test(success)
    gp, tau, tau_g2 = srs_of_the_reference(case["degree"])
    T, S, gates, r, w = case["T"], case["S"], case["Omega_gates"], case["r"], case["Omega"][1]
    mT, mS, mgates, mr, mw = limbs(T), limbs(S), limbs(gates), limbs([r])[0], limbs([w])[0]
    com_T, com_S = backend.kzg_commit_batch([mT, mS])
    t1 = P.compute_t1_T_S_zero_test(backend, mw, mT, mS)
    want_t1 = o.compute_t1_T_S_zero_test(w, T, S)
    assert ints(t1) == o.poly_trim(want_t1) and P.degree(t1) == 24
    q = P.compute_q_zero_test_from_roots(backend, mgates, t1)
    want_q = o.compute_q_zero_test_from_roots(gates, want_t1)
    assert ints(q) == o.poly_trim(want_q) and P.degree(q) == 21
    zg = backend.construct_vanishing_polynomial_from_roots(mgates)
    assert (ints(P.trim(backend.fr_poly_mul(q, zg))) == ints(t1)) == success
    proof = P.prove_T_S_zero_test(backend, mw, q, mT, mS, mr)
    want = o.prove_T_S_zero_test(gp, w, want_q, T, S, r)
    same_proof(proof, want)
    assert P.verify_T_S_zero_test(backend, tau_g2, mw, mgates, com_T, com_S, mr, proof) == success
    assert o.verify_T_S_zero_test(o.kzg_verify_with_tau(tau), w, gates, o.kzg_commit(gp, T), o.kzg_commit(gp, S), r,
                                  want) == success


def test_gadgets_without_extra_points_use_the_transform(backend):
    """deg f = k - 1: no extra points (the prover's own case, protocols.rs:589-591 with f = T): t comes from one
    inverse transform over Omega; same polynomials as the restatement's Lagrange interpolation."""
    import random
    rnd = random.Random(5)
    k = 16
    Omega = o.construct_Omega(k)
    f = [rnd.randrange(1, o.R) for _ in range(k)]
    t, t1 = P.compute_t_and_t1_product_check(backend, limbs(Omega), limbs(f))
    want_t, want_t1 = o.compute_t_and_t1_product_check(Omega, f)
    assert ints(t) == o.poly_trim(want_t) and ints(t1) == o.poly_trim(want_t1)
    t, t1 = P.compute_t_and_t1_sum_check(backend, limbs(Omega), limbs(f))
    want_t, want_t1 = o.compute_t_and_t1_sum_check(Omega, f)
    assert ints(t) == o.poly_trim(want_t) and ints(t1) == o.poly_trim(want_t1)
    # a zero of g on Omega: `inverse().unwrap()` panics in the reference (protocols.rs:349)
    g = o.poly_mul([(-Omega[3]) % o.R, 1], [rnd.randrange(o.R) for _ in range(k - 1)])
    with pytest.raises(plonk_amd.ReferencePanic):
        P.compute_t_and_t1_product_check_rational_functions(backend, limbs(Omega), limbs(f), limbs(g))
    # deg f < k - 1: `f.degree() + 1 - k` underflows in the reference
    with pytest.raises(OverflowError):
        P.compute_t_and_t1_product_check(backend, limbs(Omega), limbs(f[:5]))
