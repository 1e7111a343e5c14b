"""kzg_verify (plonk/src/common/kzg.rs:91-98) and the folded check of a whole proof, through the C ABI
(host code: runs without a GPU), against the oracle's pairing on the openings of the golden proof."""
import numpy as np
import pytest

import plonk_amd
from oracle import capi as oc
from oracle import pairing as pr
from oracle import pyref as o


@pytest.fixture(scope="module")
def setup(golden):
    return o.load_setup_json(golden["srs"])


@pytest.fixture(scope="module")
def tau_g2(golden):
    return bytes.fromhex(golden["srs"]["gp"]["tau_g2"])


@pytest.fixture(scope="module")
def openings(golden, setup):
    """(com_f, u, v, proof) of every kzg_evaluate call behind plonk/data/proof.json."""
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
    return recorded


def _arrays(checks):
    coms = oc.g1_to_limbs([c[0] for c in checks])
    us = oc.fr_to_mont([c[1] for c in checks])
    vs = oc.fr_to_mont([c[2] for c in checks])
    proofs = oc.g1_to_limbs([c[3] for c in checks])
    return coms, us, vs, proofs


def test_every_opening_of_the_golden_proof_verifies(openings, tau_g2):
    assert len(openings) == 15
    coms, us, vs, proofs = _arrays(openings)
    for i in range(len(openings)):
        assert plonk_amd.kzg_verify(tau_g2, coms[i], us[i], vs[i], proofs[i])


def test_wrong_value_point_or_proof_is_rejected(openings, tau_g2):
    coms, us, vs, proofs = _arrays(openings)
    assert not plonk_amd.kzg_verify(tau_g2, coms[2], us[2], oc.fr_to_mont([openings[2][2] + 1])[0], proofs[2])
    assert not plonk_amd.kzg_verify(tau_g2, coms[2], oc.fr_to_mont([openings[2][1] + 1])[0], vs[2], proofs[2])
    assert not plonk_amd.kzg_verify(tau_g2, coms[2], us[2], vs[2], proofs[3])
    # another polynomial's commitment (several openings share one polynomial: take one that differs)
    k = next(i for i in range(len(openings)) if not o.g1_eq(openings[i][0], openings[2][0]))
    assert not plonk_amd.kzg_verify(tau_g2, coms[k], us[2], vs[2], proofs[2])


def test_agrees_with_the_oracle_on_scaled_points(openings, tau_g2):
    """Representatives with Z != 1, the identity as a proof (a constant polynomial's quotient is empty in the
    reference, but the group element may still be handed in), and a different tau_g2."""
    com, u, v, proof = openings[5]
    z = 0x1234567
    scaled = lambda p: (p[0] * z * z % o.Q, p[1] * z * z * z % o.Q, p[2] * z % o.Q)  # noqa: E731
    coms, us, vs, proofs = _arrays([(scaled(com), u, v, scaled(proof))])
    assert plonk_amd.kzg_verify(tau_g2, coms[0], us[0], vs[0], proofs[0])
    # com = v G with the identity as proof: e(0, H) == e(identity, .) holds
    const = [(o.g1_mul(o.G1_GENERATOR, 77), 5, 77, o.G1_IDENTITY)]
    coms, us, vs, proofs = _arrays(const)
    assert plonk_amd.kzg_verify(tau_g2, coms[0], us[0], vs[0], proofs[0])
    assert pr.kzg_verify(pr.g2_decompress(tau_g2), *const[0])
    other = pr.g2_compress(pr.g2_mul(pr.G2_GEN, 12345))
    coms, us, vs, proofs = _arrays([openings[5]])
    assert not plonk_amd.kzg_verify(other, coms[0], us[0], vs[0], proofs[0])
    assert plonk_amd.kzg_verify(tau_g2, coms[0], us[0], vs[0], proofs[0])  # and back (the prepared lines are per tau_g2)


def test_invalid_tau_g2_is_refused(openings, tau_g2):
    coms, us, vs, proofs = _arrays([openings[0]])
    bad = bytearray(tau_g2)
    bad[0] &= 0x7f  # not marked compressed
    with pytest.raises(plonk_amd.ReferencePanic):
        plonk_amd.kzg_verify(bytes(bad), coms[0], us[0], vs[0], proofs[0])
    # an x with no point on the twist above it
    x = 1
    while pr.f2_sqrt(pr.f2_add(pr.f2_mul(pr.f2_mul((x, 0), (x, 0)), (x, 0)), pr.B2)) is not None:
        x += 1
    enc = bytearray((0).to_bytes(48, "big") + x.to_bytes(48, "big"))
    enc[0] |= 0x80
    with pytest.raises(plonk_amd.ReferencePanic):
        plonk_amd.kzg_verify(bytes(enc), coms[0], us[0], vs[0], proofs[0])
    # a point of the twist outside the subgroup of order r
    x = 1
    while True:
        y = pr.f2_sqrt(pr.f2_add(pr.f2_mul(pr.f2_mul((x, 0), (x, 0)), (x, 0)), pr.B2))
        if y is not None and pr.g2_mul(((x, 0), y), pr.R - 1) != pr.g2_neg(((x, 0), y)):
            break
        x += 1
    with pytest.raises(plonk_amd.ReferencePanic):
        plonk_amd.kzg_verify(pr.g2_compress(((x, 0), y)), coms[0], us[0], vs[0], proofs[0])


def test_folded_check_of_the_whole_proof(openings, tau_g2):
    coms, us, vs, proofs = _arrays(openings)
    assert plonk_amd.kzg_verify_batch(tau_g2, coms, us, vs, proofs)
    rho = np.array([[1, 0]] + [[pow(7, 5 * i + 1, 1 << 64), pow(11, 3 * i + 2, 1 << 64)] for i in range(1, 15)],
                   dtype=np.uint64)
    assert plonk_amd.kzg_verify_batch(tau_g2, coms, us, vs, proofs, rho)
    weights = [int(r[0]) | (int(r[1]) << 64) for r in rho]
    assert pr.kzg_verify_batch(pr.g2_decompress(tau_g2), openings, weights)
    for k in (0, 7, 14):
        bad = vs.copy()
        bad[k] = oc.fr_to_mont([openings[k][2] + 3])[0]
        assert not plonk_amd.kzg_verify_batch(tau_g2, coms, us, bad, proofs, rho)
    swapped = proofs.copy()
    swapped[[4, 5]] = swapped[[5, 4]]
    assert not plonk_amd.kzg_verify_batch(tau_g2, coms, us, vs, swapped)
    assert plonk_amd.kzg_verify_batch(tau_g2, coms[:0], us[:0], vs[:0], proofs[:0])


def test_tau_g2_of_kzg_setup():
    """kzg.rs:38-39: tau_g2 = G2::generator() * tau, against the oracle's G2 arithmetic and encoding."""
    for tau in (1, 2, 0x1234567, o.R - 1, 0x5d0a8f3c9b1e7a6d4c2b1a0f9e8d7c6b5a4938271605f4e3d2c1b0a99887766):
        got = plonk_amd.kzg_setup_tau_g2(oc.fr_to_mont([tau])[0])
        assert got == pr.g2_compress(pr.g2_mul(pr.G2_GEN, tau)), hex(tau)
    assert plonk_amd.kzg_setup_tau_g2(oc.fr_to_mont([0])[0]) == pr.g2_compress(None)


def test_openings_verify_under_a_fresh_srs():
    """A commitment and opening made by the oracle under another tau verify against that tau's G2 element
    produced by the library, and not against the golden one."""
    tau = 0x2f3e4d5c6b7a8990
    gp = o.kzg_setup_with_tau(8, tau)
    f = [5, 0, 7, 11, 13, 0, 0, 1, 3]
    u = 0xabcdef
    v, proof = o.kzg_evaluate(gp, f, u)
    coms, us, vs, proofs = _arrays([(o.kzg_commit(gp, f), u, v, proof)])
    tau_g2 = plonk_amd.kzg_setup_tau_g2(oc.fr_to_mont([tau])[0])
    assert plonk_amd.kzg_verify(tau_g2, coms[0], us[0], vs[0], proofs[0])
    assert not plonk_amd.kzg_verify(plonk_amd.kzg_setup_tau_g2(oc.fr_to_mont([tau + 1])[0]), coms[0], us[0], vs[0],
                                    proofs[0])


def test_kzg_verify_like_the_reference_test():
    """plonk/tests/kzg_tests.rs:148-197: com_f = [f(tau)]G and proof = [q(tau)]G, q = (f - v) / (X - u), computed from
    the trapdoor directly; kzg_verify must return true."""
    import random
    rnd = random.Random(148)
    tau = rnd.randrange(1, o.R)
    f = [rnd.randrange(o.R) for _ in range(11)]  # degree 10
    u = rnd.randrange(o.R)
    v = o.poly_eval(f, u)
    q, rem = o.poly_divmod(o.poly_sub(f, [v]), [(-u) % o.R, 1])
    assert o.poly_trim(rem) == []
    com_f = o.g1_mul(o.G1_GENERATOR, o.poly_eval(f, tau))
    proof = o.g1_mul(o.G1_GENERATOR, o.poly_eval(q, tau))
    coms, us, vs, proofs = _arrays([(com_f, u, v, proof)])
    tau_g2 = plonk_amd.kzg_setup_tau_g2(oc.fr_to_mont([tau])[0])
    assert plonk_amd.kzg_verify(tau_g2, coms[0], us[0], vs[0], proofs[0])


def test_points_off_the_curve_are_refused(openings, tau_g2):
    """The reference's G1 values are on the curve by construction (validated deserialisation); raw limbs are
    not, and a pairing of a point off the curve means nothing: PLONK_HIP_ERR_INVALID_POINT, one by one and
    inside a fold, for the commitment and for the proof; a Z != 1 representative of a good point still passes."""
    coms, us, vs, proofs = _arrays(openings)
    bad = coms[4].copy()
    bad[6] ^= np.uint64(1)  # one bit of Y
    with pytest.raises(plonk_amd.ReferencePanic) as e:
        plonk_amd.kzg_verify(tau_g2, bad, us[4], vs[4], proofs[4])
    assert e.value.code == -9
    bad_proof = proofs[4].copy()
    bad_proof[0] ^= np.uint64(2)  # one bit of X
    with pytest.raises(plonk_amd.ReferencePanic):
        plonk_amd.kzg_verify(tau_g2, coms[4], us[4], vs[4], bad_proof)
    folded = coms.copy()
    folded[9] = bad
    with pytest.raises(plonk_amd.ReferencePanic):
        plonk_amd.kzg_verify_batch(tau_g2, folded, us, vs, proofs)
    assert plonk_amd.kzg_verify_batch(tau_g2, coms, us, vs, proofs)
