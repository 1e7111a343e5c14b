"""Pins oracle/pyref.py on every golden vector and known-answer test the reference holds for
the KZG / polynomial path (plonk/data/*.json; plonk/tests/{polynomials,utils,kzg,field}_tests.rs)."""
import random

import pytest

from oracle import pyref as o


@pytest.fixture(scope="module")
def setup(golden):
    return o.load_setup_json(golden["srs"])


def test_srs_points_decode_and_reencode(golden, setup):
    # plonk/data/srs.json:8 is the standard generator encoding
    assert o.g1_eq(setup.gp.tau_powers_g1[0], o.G1_GENERATOR)
    for hexstr, p in zip(golden["srs"]["gp"]["tau_powers_g1"], setup.gp.tau_powers_g1):
        assert o.g1_on_curve(p)
        assert o.g1_hex(p) == hexstr
    assert len(setup.gp.tau_powers_g1) == 22 and setup.d == 12


def test_fr_modulus_minus_one_limbs():
    # plonk/tests/field_tests.rs:7-17
    limbs = [0xFFFFFFFF00000000, 0x53BDA402FFFE5BFE, 0x3339D80809A1D805, 0x73EDA753299D7D48]
    assert sum(l << (64 * i) for i, l in enumerate(limbs)) == o.R - 1


def test_proving_key_reproduced(golden, setup):
    assert o.reproduce_proving_key(setup) == golden["proving_key"]


def test_verification_key_reproduced(golden, setup):
    assert o.reproduce_verification_key(setup) == golden["verification_key"]


def test_proof_reproduced_byte_for_byte(golden, setup):
    proof = o.reproduce_proof(setup, golden["proving_key"], golden["verification_key"])
    assert proof == golden["proof"]


# ---- plonk/tests/polynomials_tests.rs
def test_poly_kats():
    m = o.R
    assert o.poly_add([1, 2, 3], [4, 5, 6]) == [5, 7, 9]                      # :21-31
    assert o.poly_mul([1, 1], [1, 1]) == [1, 2, 1]                            # :33-43
    assert o.poly_divmod([m - 1, 0, 1], [m - 1, 1])[0] == [1, 1]              # :45-55
    q, r = o.poly_divmod([4, m - 5, 0, 0, 0, 0, 0, 0, 0, 1], [1, 0, 1])       # :57-92
    assert q == [0, m - 1, 0, 1, 0, m - 1, 0, 1] and r == [4, m - 4]
    assert o.poly_eval([0, 1, 1], 5) == 30                                    # :152-158
    assert o.interpolate_polynomial([0, 1, 2], [0, 1, 8]) == [0, m - 2, 3]    # :160-177
    assert o.poly_pow([1, 1], 2) == [1, 2, 1] and o.poly_pow([1, 1], 3) == [1, 3, 3, 1]  # :200-215
    assert o.compose_polynomials([0, 1, 1], [1, 1]) == [2, 3, 1]              # :217-234


def test_product_of_monomials_over_subgroup():
    # plonk/tests/polynomials_tests.rs:120-150 (order reduced from 1024 to 64 for CPU time)
    n = 64
    g = pow(o.FR_GENERATOR, (o.R - 1) >> 6, o.R)
    prod = [1]
    for i in range(n):
        prod = o.poly_mul(prod, [(-pow(g, i, o.R)) % o.R, 1])
    assert prod == o.construct_vanishing_polynomial(n)


def test_random_division_and_interpolation():
    rnd = random.Random(7)
    for _ in range(10):  # :94-118
        a = o.poly_trim([rnd.randrange(o.R) for _ in range(rnd.randrange(1, 52))])
        b = o.poly_trim([rnd.randrange(o.R) for _ in range(rnd.randrange(1, 52))])
        q, r = o.poly_divmod(a, b)
        assert o.poly_add(r, o.poly_mul(q, b)) == a
        assert o.poly_degree(r) < o.poly_degree(b) or (o.poly_degree(r) == 0 and o.poly_degree(b) == 0)
    for _ in range(4):  # :179-198
        deg = rnd.randrange(0, 40)
        p = [rnd.randrange(o.R) for _ in range(deg)] + [rnd.randrange(1, o.R)]
        xs = rnd.sample(range(1, 1 << 62), deg + 1)
        ys = [o.poly_eval(p, x) for x in xs]
        assert o.interpolate_polynomial(xs, ys) == p


# ---- plonk/tests/utils_tests.rs
def test_omega_and_vanishing():
    for k in (2, 3, 4, 12, 16):  # :12-22
        Om = o.construct_Omega(k)
        assert len(Om) == k and Om[0] == 1 and pow(Om[1], k, o.R) == 1
        assert len(set(Om)) == k
        Z = o.construct_vanishing_polynomial(k)  # :24-38
        assert all(o.poly_eval(Z, x) == 0 for x in Om)
        assert o.construct_vanishing_polynomial_from_roots(Om) == Z  # :40-52


def test_challenge_derivation_shape(setup):
    # :54-134 -- re-derivation from the same commitments gives the same element; domain
    # separated challenges are pairwise distinct
    pts = setup.gp.tau_powers_g1[:2]
    r1 = o.derive_challenge_from_commitments(pts)
    assert r1 == o.derive_challenge_from_commitments(list(pts)) and 0 <= r1 < o.R
    ch = o.derive_multiple_challenges_from_commitments(pts, 3)
    assert len(set(ch)) == 3 and ch == o.derive_multiple_challenges_from_commitments(pts, 3)


# ---- plonk/tests/kzg_tests.rs
def test_kzg_commit_and_eval_known_answers():
    rnd = random.Random(11)
    degree = 10
    tau = rnd.randrange(1, o.R)
    gp = o.kzg_setup_with_tau(degree, tau)
    f = [rnd.randrange(o.R) for _ in range(degree)] + [rnd.randrange(1, o.R)]
    com = o.kzg_commit(gp, f)  # :36-69
    assert o.g1_eq(com, o.g1_mul(o.G1_GENERATOR, o.poly_eval(f, tau)))
    u = rnd.randrange(o.R)  # :95-146
    v, proof = o.kzg_evaluate(gp, f, u)
    q, _ = o.poly_divmod(o.poly_sub(f, [v]), [(-u) % o.R, 1])
    assert v == o.poly_eval(f, u)
    assert o.g1_eq(proof, o.g1_mul(o.G1_GENERATOR, o.poly_eval(q, tau)))
    with pytest.raises(o.CommitError):  # :71-93
        o.kzg_commit(gp, f + [1])


def test_transforms_match_reference_definitions():
    rnd = random.Random(5)
    for n in (2, 4, 8, 32):
        co = [rnd.randrange(o.R) for _ in range(n)]
        ev = o.ntt(co)
        Om = o.construct_Omega(n)
        assert ev == [o.poly_eval(co, x) for x in Om]          # protocols.rs:243-250
        assert o.interpolate_polynomial(Om, ev) == o.poly_trim(co)  # polynomials.rs:50-58
        assert o.ntt(ev, invert=True) == co


def test_example_as_a_circuit_description_reproduces_the_golden_files(golden):
    setup = o.load_setup_json(golden["srs"])
    ex = o.example_circuit()
    assert o.reproduce_proving_key(setup, ex) == golden["proving_key"]
    assert o.reproduce_verification_key(setup, ex) == golden["verification_key"]
    assert o.reproduce_proof(setup, golden["proving_key"], golden["verification_key"], ex) == golden["proof"]


def test_synthetic_circuit_family_proves():
    # 7 gates, d = 24: the prover's own assertions (gate constraint, wiring, divisibility, output) hold
    circuit = o.synthetic_circuit(3, 5)
    d = len(circuit.trace)
    assert d == 24 and len(circuit.gates) == 7 and sorted(circuit.wiring) == list(range(d))
    for l, kind in enumerate(circuit.gates):
        a, b, out = circuit.trace[3 * l:3 * l + 3]
        assert out == ((a + b) if kind else a * b) % o.R
    setup = o.Setup(7, 2, 1, d, o.kzg_setup_with_tau(2 * d - 3, 0x1234567))
    pk, vk = o.reproduce_proving_key(setup, circuit), o.reproduce_verification_key(setup, circuit)
    proof = o.reproduce_proof(setup, pk, vk, circuit)
    assert proof["output"] == o.fr_to_string(circuit.output)
    assert len(proof["proof_T_W_prescribed_permutation"]) == 16
