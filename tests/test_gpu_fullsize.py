"""The benchmark configurations at their full sizes, checked through size-independent properties:

* 16 coset quotients 2^21 -> 2^20 coefficients (`compute_q_zero_test`, plonk/src/common/protocols.rs:157-166)
  on f = q * (X^(2^20) - 1) with q uniform: the quotient must come back exactly; one f that X^k - 1 does not
  divide must be flagged, and the exact long-division entry point must still return q for it;
* 16 transforms of 2^20 points: forward then inverse restores the input, and polynomial 0 equals the
  oracle's transform;
* the whole prover for a synthetic circuit of 2^20 - 1 gates (plonk/src/prover/mod.rs:18-74): every
  commitment and opening equals the trapdoor evaluation of the test SRS and the verifier's field
  equations hold."""
import numpy as np
import pytest

from oracle import capi as c
from tests.gpu_util import backend  # noqa: F401

pytestmark = pytest.mark.gpu

R_LIMBS = [0xFFFFFFFF00000001, 0x53BDA402FFFE5BFE, 0x3339D80809A1D805, 0x73EDA753299D7D48]


def rand(rng, n):
    a = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    return a


def neg_mod_r(x):
    """r - x limb by limb (x below r; 0 stays 0): the additive inverse, whatever the Montgomery radix."""
    out = np.empty_like(x)
    borrow = np.zeros(x.shape[0], dtype=bool)
    for i in range(4):
        ri = np.uint64(R_LIMBS[i])
        xi = x[:, i]
        out[:, i] = ri - xi - borrow.astype(np.uint64)  # wraps modulo 2^64 as intended
        borrow = (xi > ri) | ((xi == ri) & borrow)
    out[~x.any(axis=1)] = 0
    return out


def test_neg_helper_against_the_oracle():
    rng = np.random.default_rng(1)
    x = rand(rng, 64)
    x[3] = 0
    assert np.array_equal(neg_mod_r(x), c.fr_sub(np.zeros_like(x), x))


def test_sixteen_coset_quotients_of_2_to_21_coefficients(backend):
    log_k, batch = 20, 16
    k = 1 << log_k
    rng = np.random.default_rng(2020)
    q = rand(rng, batch * k).reshape(batch, k, 4)
    f = np.empty((batch, 2 * k, 4), dtype=np.uint64)
    f[:, :k] = neg_mod_r(q.reshape(-1, 4)).reshape(batch, k, 4)  # q * (X^k - 1) = q X^k - q
    f[:, k:] = q
    got, divisible = backend.quotient_coset(f)
    assert all(divisible)
    assert np.array_equal(got, q)
    # X^k - 1 does not divide f + 1: the coset path says so, long division still yields q (remainder 1)
    bad = f[5].copy()
    bad[0] = c.fr_add(bad[0:1], c.fr_to_mont([1]))[0]
    got_bad, divisible_bad = backend.quotient_coset(bad)
    assert divisible_bad == [False]
    assert np.array_equal(backend.compute_q_zero_test(k, bad), q[5])


def test_sixteen_transforms_of_2_to_20_points(backend):
    log_n, batch = 20, 16
    n = 1 << log_n
    rng = np.random.default_rng(77)
    data = rand(rng, batch * n)
    d = backend.to_device(data)
    try:
        backend.fr_ntt_dev(d, log_n, batch, False)
        backend.sync()
        fwd = backend.from_device(d, (batch * n, 4))
        assert np.array_equal(fwd[:n], c.fr_ntt(data[:n]))
        backend.fr_ntt_dev(d, log_n, batch, True)
        backend.sync()
        assert np.array_equal(backend.from_device(d, (batch * n, 4)), data)
    finally:
        backend.free(d)


def test_proof_of_a_circuit_of_2_to_20_gates():
    import plonk_amd
    from plonk_amd import prover as pv

    be = plonk_amd.Backend(device=0)
    try:
        circuit = pv.synthetic_circuit(20, 7)
        p = pv.Prover(be, circuit.d)
        tau = 0x123456789ABCDEF0123
        p.setup(tau)
        keys = p.keys(circuit)
        p.kzg_log = []
        proof = p.prove_resident(circuit, keys)
        assert p.audit_with_tau(tau) >= 2  # every logged commitment / opening equals [f(tau)]G / [q(tau)]G
        assert p.check_identities((circuit.number_gates, circuit.pub_inputs), proof, keys)
        # the openings by pairings, folded into one check (public parameters only)
        tau_g2 = plonk_amd.kzg_setup_tau_g2(pv.to_mont([tau])[0])
        assert p.verify_openings(tau_g2)
        # and the whole verifier on the proof as it is serialised (verifier::run, plonk/src/verifier/mod.rs:13-42)
        from plonk_amd import verifier
        vk = {"com_S": p.hex_of(keys["com_S"]), "com_W": p.hex_of(keys["com_W"])}
        v = verifier.Verifier(be, circuit.d, circuit.number_gates, len(circuit.pub_inputs), tau_g2)
        assert v.run(vk, proof)
        forged = dict(proof, output=str((int(proof["output"]) + 1) % pv.R_MOD))
        assert not v.run(vk, forged)
        p.kzg_log = None
        assert p.prove_resident(circuit, keys) == proof  # deterministic
    finally:
        be.close()
