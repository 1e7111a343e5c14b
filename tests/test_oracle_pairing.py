"""Pins oracle/pairing.py (the restated pairing check of kzg_verify, plonk/src/common/kzg.rs:91-98) on the
reference's own data: the golden SRS relates tau G and tau H consistently, every opening of the golden proof
verifies, tampered ones do not, and the folded check agrees."""
import pytest

from oracle import pairing as pr
from oracle import pyref as o


@pytest.fixture(scope="module")
def setup(golden):
    return o.load_setup_json(golden["srs"])


@pytest.fixture(scope="module")
def tau_g2(golden):
    return pr.g2_decompress(bytes.fromhex(golden["srs"]["gp"]["tau_g2"]))


@pytest.fixture(scope="module")
def openings(golden, setup):
    """(com_f, u, v, proof) of every kzg_evaluate call of the golden proof, recorded while the oracle reproduces
    plonk/data/proof.json byte for byte."""
    recorded = []
    original = o.kzg_evaluate

    def recording(gp, f, u):
        v, proof = original(gp, f, u)
        recorded.append((o.kzg_commit(gp, f), u % o.R, v % o.R, proof))
        return v, proof

    o.kzg_evaluate = recording
    try:
        assert o.reproduce_proof(setup, golden["proving_key"], golden["verification_key"]) == golden["proof"]
    finally:
        o.kzg_evaluate = original
    assert len(recorded) == 15  # 2 + 5 + 7 + 1 (protocols.rs:229-231, 804-808, 706-718, verifier/part4.rs:20-29)
    return recorded


def test_generators():
    assert pr.g2_on_curve(pr.G2_GEN) and pr.g2_mul(pr.G2_GEN, pr.R - 1) == pr.g2_neg(pr.G2_GEN)
    assert (pr.G1_GEN[0], pr.G1_GEN[1], 1) == o.G1_GENERATOR


def test_golden_tau_g2_decodes_and_matches_tau_g1(golden, setup, tau_g2):
    assert pr.g2_on_curve(tau_g2)
    assert pr.g2_compress(tau_g2).hex() == golden["srs"]["gp"]["tau_g2"]
    tau_g1 = o.g1_to_affine(setup.gp.tau_powers_g1[1])
    minus_g1 = o.g1_to_affine(o.g1_neg(o.G1_GENERATOR))
    # e(tau G, H) = e(G, tau H): holds only if H is the generator the reference multiplied by tau (kzg.rs:39)
    assert pr.pairing_product_is_one([(tau_g1, pr.G2_GEN), (minus_g1, tau_g2)])
    assert not pr.pairing_product_is_one([(tau_g1, pr.G2_GEN), (minus_g1, pr.g2_double(tau_g2))])


def test_bilinear_and_non_degenerate():
    a, b = 0x1234567, 0x89abcdef01
    aG = o.g1_to_affine(o.g1_mul(o.G1_GENERATOR, a))
    abG = o.g1_to_affine(o.g1_neg(o.g1_mul(o.G1_GENERATOR, a * b)))
    bH = pr.g2_mul(pr.G2_GEN, b)
    assert pr.pairing_product_is_one([(aG, bH), (abG, pr.G2_GEN)])
    assert pr.pairing(pr.G1_GEN, pr.G2_GEN) != pr.F12_ONE
    assert pr.pairing(None, pr.G2_GEN) == pr.F12_ONE and pr.pairing(pr.G1_GEN, None) == pr.F12_ONE


def test_openings_of_the_golden_proof_verify(openings, tau_g2):
    for k in (0, 6, 14):
        com, u, v, proof = openings[k]
        assert pr.kzg_verify(tau_g2, com, u, v, proof)
    com, u, v, proof = openings[3]
    assert not pr.kzg_verify(tau_g2, com, u, (v + 1) % o.R, proof)
    assert not pr.kzg_verify(tau_g2, com, u, v, o.g1_double(proof))


def test_folded_check_agrees(openings, tau_g2):
    rho = [1] + [pow(7, 3 * i + 1, o.R) for i in range(1, len(openings))]
    assert pr.kzg_verify_batch(tau_g2, openings, rho)
    bad = list(openings)
    com, u, v, proof = bad[9]
    bad[9] = (com, u, (v + 5) % o.R, proof)
    assert not pr.kzg_verify_batch(tau_g2, bad, rho)
