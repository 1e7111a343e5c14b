"""GPU parity of the gadget provers off `prover::run`'s path -- equality, zero test, zero on roots, product
check, sum check, product check of rational functions (plonk/src/common/protocols.rs:126-556) -- end to end
through the C ABI (plonk_amd/protocols.py) against the oracle's restatement, on the reference's own test inputs
regenerated draw by draw (plonk/tests/protocols_tests.rs:24-644: k = 8, deg f = 10, so compute_t_and_t1_* draw
three extra interpolation points from ark_std::test_rng()).  Bit-exact: every polynomial coefficient, opened
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
