"""The restated `ark_std::test_rng()` / `Fr::rand` (oracle/arkrng.py) pinned on the reference's own data, and
the gadget branch it unlocks: with the reference's inputs regenerated draw by draw
(plonk/tests/protocols_tests.rs), every success case proves and verifies and every fail case is rejected,
as the reference's tests assert."""
import numpy as np
import pytest

from oracle import arkrng as a
from oracle import pyref as o
from tests import protocol_flows as flows


def test_tau_drawn_like_the_reference_reproduces_the_golden_srs(golden):
    # plonk/src/common/kzg.rs:25-36 against plonk/data/srs.json:8-29: all 22 points
    gp, tau = a.kzg_setup(21)
    want = o.load_setup_json(golden["srs"]).gp.tau_powers_g1
    assert [o.g1_hex(p) for p in gp.tau_powers_g1] == [o.g1_hex(p) for p in want]
    # and tau_g2 = [tau]H (kzg.rs:38-39)
    from oracle import pairing
    assert pairing.g2_compress(pairing.g2_mul(pairing.G2_GEN, tau)).hex() == golden["srs"]["gp"]["tau_g2"]


def test_other_round_counts_and_readings_do_not():
    want = o.g1_hex(a.kzg_setup(1)[0].tau_powers_g1[1])
    for rounds in (8, 20):
        v = a.fr_rand_montgomery_limbs(a.StdRng(a.TEST_RNG_SEED, rounds))
        assert o.g1_hex(o.g1_mul(o.G1_GENERATOR, v * a.R_INV % o.R)) != want
    v = a.fr_rand_montgomery_limbs(a.test_rng())
    assert o.g1_hex(o.g1_mul(o.G1_GENERATOR, v)) != want  # limbs read as the canonical integer


def test_product_side_generator_draws_the_same_limbs():
    from plonk_amd.ark_rng import TestRng
    mine, theirs = TestRng(), a.test_rng()
    for _ in range(40):
        limbs = mine.fr()
        assert sum(int(x) << (64 * i) for i, x in enumerate(limbs)) == a.fr_rand_montgomery_limbs(theirs)


def test_chacha_block_known_answer():
    # RFC 8439 section 2.3.2 (20 rounds, 32-bit counter = 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00):
    # in the 64-bit-counter layout the same state is counter = 1 | 0x09000000 << 32, stream = 0x4a000000
    key = [int.from_bytes(bytes(range(4 * i, 4 * i + 4)), "little") for i in range(8)]
    out = a.chacha_block(key, 1 | (0x09000000 << 32), 0x4A000000, 20)
    assert out[:4] == [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3]
    assert out[12:] == [0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]


@pytest.mark.parametrize("success", [True, False])
def test_gadgets_with_extra_random_points(success):
    """protocols_tests.rs:253-644 on the oracle: k = 8, deg f = 10, three extra interpolation points drawn
    inside compute_t_and_t1_* (protocols.rs:251-253, 354-356, 469-471)."""
    for name in ("product_check", "sum_check", "rational_check"):
        c = getattr(flows, name)(success)
        gp, tau = a.kzg_setup(c["degree"])
        check = o.kzg_verify_with_tau(tau)
        Omega, f, r, w = c["Omega"], c["f"], c["r"], c["Omega"][1]
        assert o.poly_degree(f) == 10
        com_f = o.kzg_commit(gp, f)
        if name == "product_check":
            t, t1 = o.compute_t_and_t1_product_check(Omega, f)
        elif name == "sum_check":
            t, t1 = o.compute_t_and_t1_sum_check(Omega, f)
        else:
            t, t1 = o.compute_t_and_t1_product_check_rational_functions(Omega, f, c["g"])
        assert o.poly_degree(t) == 10  # eight points of Omega and three drawn ones
        q = o.compute_q_zero_test(flows.K, t1)
        divisible = o.poly_mul(q, o.construct_vanishing_polynomial(flows.K)) == o.poly_trim(t1)
        # the rational "success" case of the reference has g = f: t1 = 0 there, trivially divisible
        assert divisible == success
        if name == "product_check":
            p = o.prove_product_check(gp, w, flows.K, t, q, f, r)
            ok = o.verify_product_check(check, w, flows.K, com_f, r, p)
        elif name == "sum_check":
            p = o.prove_sum_check(gp, w, flows.K, t, q, f, r)
            ok = o.verify_sum_check(check, w, flows.K, com_f, r, p)
        else:
            com_g = o.kzg_commit(gp, c["g"])
            p = o.prove_product_check_rational_functions(gp, w, flows.K, t, q, f, c["g"], r)
            ok = o.verify_product_check_rational_functions(check, w, flows.K, com_f, com_g, r, p)
        assert ok == success, name


@pytest.mark.parametrize("success", [True, False])
def test_equality_and_zero_tests(success):
    """protocols_tests.rs:24-251 on the oracle."""
    c = flows.equality(success)
    gp, tau = a.kzg_setup(c["degree"])
    check = o.kzg_verify_with_tau(tau)
    p = o.prove_equality(gp, c["f"], c["g"], c["r"])
    assert o.verify_equality(check, o.kzg_commit(gp, c["f"]), o.kzg_commit(gp, c["g"]), c["r"], p) == success
    c = flows.zero_test(success)
    q = o.compute_q_zero_test(flows.K, c["f"])
    p = o.prove_zero_test(gp, c["f"], q, c["r"])
    assert o.verify_zero_test(check, flows.K, o.kzg_commit(gp, c["f"]), c["r"], p) == success
    c = flows.zero_on_roots(success)
    q = o.compute_q_zero_test_from_roots(c["roots"], c["f"])
    p = o.prove_zero_test(gp, c["f"], q, c["r"])
    assert o.verify_zero_on_roots_test(check, c["roots"], o.kzg_commit(gp, c["f"]), c["r"], p) == success


@pytest.mark.parametrize("success", [True, False])
def test_permutation_prescribed_permutation_and_T_S_gadgets(success):
    """protocols_tests.rs:646-1149 on the oracle: permutation check (through the product check of rational
    functions), prescribed permutation check, T_S zero test; the degrees the reference asserts (:684-685, :829-831,
    :1037, :1058) and its verdicts."""
    # :646-776
    c = flows.permutation_check(success)
    gp, tau = a.kzg_setup(c["degree"])
    check = o.kzg_verify_with_tau(tau)
    Omega, f, g, r, w = c["Omega"], c["f"], c["g"], c["r"], c["Omega"][1]
    assert o.poly_degree(f) == 10 and o.poly_degree(g) == 10
    t, t1 = o.compute_t_and_t1_product_check_rational_functions(Omega, f, g)
    q = o.compute_q_zero_test(flows.K, t1)
    assert (o.poly_mul(q, o.construct_vanishing_polynomial(flows.K)) == o.poly_trim(t1)) == success
    p = o.prove_product_check_rational_functions(gp, w, flows.K, t, q, f, g, r)
    assert o.verify_product_check_rational_functions(check, w, flows.K, o.kzg_commit(gp, f), o.kzg_commit(gp, g), r,
                                                     p) == success
    # :778-943
    c = flows.prescribed_permutation_check(success)
    f, g, W = c["f"], c["g"], c["W"]
    assert [o.poly_degree(x) for x in (f, g, W)] == [10, 10, 10]
    assert all(o.poly_eval(f, x) == o.poly_eval(g, o.poly_eval(W, x)) for x in Omega) == success
    t, t1 = o.compute_t_and_t1_prescribed_permutation_check(Omega, f, g, W, c["r"], c["s"])
    q = o.compute_q_zero_test(flows.K, t1)
    assert (o.poly_mul(q, o.construct_vanishing_polynomial(flows.K)) == o.poly_trim(t1)) == success
    p = o.prove_prescribed_permutation_check(gp, w, flows.K, t, q, f, g, W, c["rp"])
    assert o.verify_prescribed_permutation_check(check, w, flows.K, o.kzg_commit(gp, f), o.kzg_commit(gp, g),
                                                 o.kzg_commit(gp, W), c["r"], c["s"], c["rp"], p) == success
    # :945-1149
    c = flows.t_s_zero_test(success)
    gp, tau = a.kzg_setup(c["degree"])
    check = o.kzg_verify_with_tau(tau)
    T, S, gates, w = c["T"], c["S"], c["Omega_gates"], c["Omega"][1]
    assert o.poly_degree(T) == 11 and o.poly_degree(S) == 2
    t1 = o.compute_t1_T_S_zero_test(w, T, S)
    assert o.poly_degree(t1) == 24
    assert all((o.poly_eval(t1, y) == 0) == success for y in gates)
    q = o.compute_q_zero_test_from_roots(gates, t1)
    assert o.poly_degree(q) == 21
    z = o.construct_vanishing_polynomial_from_roots(gates)
    assert (o.poly_mul(q, z) == o.poly_trim(t1)) == success
    p = o.prove_T_S_zero_test(gp, w, q, T, S, c["r"])
    assert o.verify_T_S_zero_test(check, w, gates, o.kzg_commit(gp, T), o.kzg_commit(gp, S), c["r"], p) == success


def test_shuffle_is_a_permutation_drawn_in_32_bit_words():
    rng, ref = a.test_rng(), a.test_rng()
    items = list(range(8))
    a.shuffle(rng, items)
    assert sorted(items) == list(range(8)) and items != list(range(8))
    # seven swaps, each one accepted 32-bit draw after any rejected ones (the zone test accepts at least half)
    words = [ref.next_u32() for _ in range(64)]
    follow = rng.next_u32()
    assert any(follow == words[n] for n in range(7, 64))
    assert a.gen_range_u32(a.test_rng(), 1) == 0


def test_one_pairing_check_agrees_with_the_trapdoor_check():
    from oracle import pairing
    c = flows.sum_check(True)
    gp, tau = a.kzg_setup(c["degree"])
    tau_g2 = pairing.g2_mul(pairing.G2_GEN, tau)
    v, proof = o.kzg_evaluate(gp, c["f"], c["r"])
    com = o.kzg_commit(gp, c["f"])
    assert pairing.kzg_verify(tau_g2, com, c["r"], v, proof) and o.kzg_verify_with_tau(tau)(com, c["r"], v, proof)
    assert not pairing.kzg_verify(tau_g2, com, c["r"], (v + 1) % o.R, proof)
    assert not o.kzg_verify_with_tau(tau)(com, c["r"], (v + 1) % o.R, proof)
