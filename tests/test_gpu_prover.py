"""End-to-end: the example circuit's proof with every commitment and opening computed on the
GPU through the C ABI must equal the reference's plonk/data/proof.json byte for byte."""
import numpy as np
import pytest

from oracle import capi as c
from oracle import pyref as o
from tests.gpu_util import backend  # noqa: F401

pytestmark = pytest.mark.gpu


def test_example_proof_bytes_through_gpu(backend, golden, monkeypatch):
    setup = o.load_setup_json(golden["srs"])
    backend.srs_load_projective(c.g1_to_limbs(setup.gp.tau_powers_g1))
    calls = {"commit": 0, "open": 0}

    def gpu_commit(gp, f):
        f = o.poly_trim(f)
        calls["commit"] += 1
        return c.g1_from_limbs(backend.kzg_commit(c.fr_to_mont(f)))[0]

    def gpu_evaluate(gp, f, u):
        f = o.poly_trim(f)
        calls["open"] += 1
        v, proof = backend.kzg_evaluate(c.fr_to_mont(f), c.fr_to_mont([u])[0])
        return c.fr_from_mont(v)[0], c.g1_from_limbs(proof)[0]

    monkeypatch.setattr(o, "kzg_commit", gpu_commit)
    monkeypatch.setattr(o, "kzg_evaluate", gpu_evaluate)
    assert o.reproduce_verification_key(setup) == golden["verification_key"]
    proof = o.reproduce_proof(setup, golden["proving_key"], golden["verification_key"])
    assert proof == golden["proof"]
    assert calls == {"commit": 2 + 6, "open": 15}


def _polynomial_layer_on_gpu(backend, monkeypatch):
    """Replaces the restatement's commitments, openings, interpolation, f(wX), products and both
    quotients (common/kzg.rs, common/polynomials.rs, common/protocols.rs) by calls through the C
    ABI; the gadget logic, the challenges and the JSON stay with the restatement.  Returns the call
    counters."""
    calls = {k: 0 for k in ("commit", "open", "interpolate_omega", "interpolate", "shift", "mul", "div_vanishing",
                            "div_roots")}
    M, U = c.fr_to_mont, c.fr_from_mont
    cpu_compose = o.compose_polynomials

    def gpu_commit(gp, f):
        calls["commit"] += 1
        return c.g1_from_limbs(backend.kzg_commit(M(o.poly_trim(f))))[0]

    def gpu_evaluate(gp, f, u):
        calls["open"] += 1
        v, proof = backend.kzg_evaluate(M(o.poly_trim(f)), M([u])[0])
        return U(v)[0], c.g1_from_limbs(proof)[0]

    def gpu_interpolate(xs, ys):
        n = len(xs)
        if n > 1 and (o.R - 1) % n == 0 and list(xs) == o.construct_Omega(n):
            calls["interpolate_omega"] += 1
            return o.poly_trim(U(backend.fr_ntt_mixed(M(ys), inverse=True)))
        calls["interpolate"] += 1
        return o.poly_trim(U(backend.interpolate_polynomial(M(xs), M(ys))))

    def gpu_compose(f, g):
        g = list(g)
        if len(g) == 2 and g[0] == 0 and len(f) >= 2:
            calls["shift"] += 1
            return o.poly_trim(U(backend.compose_with_scaling(M(f), M([g[1]])[0])))
        return cpu_compose(f, g)  # f of degree 1 (r - s W): two coefficients, stays on the host

    def gpu_mul(a, b):
        if not a or not b:
            return []
        calls["mul"] += 1
        return o.poly_trim(U(backend.fr_poly_mul(M(a), M(b))))

    def gpu_q_vanishing(k, f):
        calls["div_vanishing"] += 1
        return o.poly_trim(U(backend.compute_q_zero_test(k, M(o.poly_trim(f)))))

    def gpu_q_roots(roots, f):
        calls["div_roots"] += 1
        return o.poly_trim(U(backend.compute_q_zero_test_from_roots(M(roots), M(o.poly_trim(f)))))

    monkeypatch.setattr(o, "kzg_commit", gpu_commit)
    monkeypatch.setattr(o, "kzg_evaluate", gpu_evaluate)
    monkeypatch.setattr(o, "interpolate_polynomial", gpu_interpolate)
    monkeypatch.setattr(o, "compose_polynomials", gpu_compose)
    monkeypatch.setattr(o, "poly_mul", gpu_mul)
    monkeypatch.setattr(o, "compute_q_zero_test", gpu_q_vanishing)
    monkeypatch.setattr(o, "compute_q_zero_test_from_roots", gpu_q_roots)
    return calls


def test_example_proof_bytes_with_the_polynomial_layer_on_the_gpu(backend, golden, monkeypatch):
    """The same proof, now with interpolation, f(wX), products and both quotients of
    common/polynomials.rs and common/protocols.rs also computed on the GPU."""
    setup = o.load_setup_json(golden["srs"])
    backend.srs_load_projective(c.g1_to_limbs(setup.gp.tau_powers_g1))
    calls = _polynomial_layer_on_gpu(backend, monkeypatch)
    assert o.reproduce_proving_key(setup) == golden["proving_key"]
    assert o.reproduce_verification_key(setup) == golden["verification_key"]
    proof = o.reproduce_proof(setup, golden["proving_key"], golden["verification_key"])
    assert proof == golden["proof"]
    assert calls["commit"] == 8 and calls["open"] == 15
    assert min(calls.values()) > 0, calls


def _synthetic_setup(backend, log_gates, seed):
    circuit = o.synthetic_circuit(log_gates, seed)
    d = len(circuit.trace)
    tau = 0x0123456789ABCDEF0FEDCBA987654321 + seed
    # the quotient of the gate check has degree 2d - 3 (setup_global_params/mod.rs:60-61: 21 for d = 12)
    pts = backend.kzg_setup(c.fr_to_mont([tau])[0], 2 * d - 3)
    return circuit, d, tau, pts


@pytest.mark.parametrize("log_gates", [3, 5])
def test_synthetic_circuit_proof_matches_the_restatement(backend, monkeypatch, log_gates):
    """Larger circuits of the example's family (2^k - 1 gates, d = 3 * 2^k trace rows): the whole
    proof computed through the C ABI equals the restatement's, field by field."""
    circuit, d, tau, pts = _synthetic_setup(backend, log_gates, 11)
    setup = o.Setup((1 << log_gates) - 1, 2, 1, d, o.GlobalParameters(c.g1_from_limbs(pts), b""))
    pk = o.reproduce_proving_key(setup, circuit)
    vk = o.reproduce_verification_key(setup, circuit)
    want = o.reproduce_proof(setup, pk, vk, circuit)
    calls = _polynomial_layer_on_gpu(backend, monkeypatch)
    assert o.reproduce_proving_key(setup, circuit) == pk
    assert o.reproduce_verification_key(setup, circuit) == vk
    assert o.reproduce_proof(setup, pk, vk, circuit) == want
    assert calls["commit"] == 8 and calls["open"] == 15 and min(calls.values()) > 0, calls


def test_synthetic_circuit_of_255_gates(backend, monkeypatch):
    """d = 768 rows, beyond what the quadratic and cubic loops of the restatement finish quickly: the
    GPU-backed prover alone, checked by the prover's own assertions (the gate constraint vanishes on
    the gate rows, the wiring holds on Omega, t1 = q * Z_Omega exactly, the opening at the last gate
    gives the output) and by commit == [f(tau)]G for the trace."""
    log_gates = 8
    circuit, d, tau, pts = _synthetic_setup(backend, log_gates, 12)
    setup = o.Setup((1 << log_gates) - 1, 2, 1, d, o.GlobalParameters([None] * (2 * d - 2), b""))
    calls = _polynomial_layer_on_gpu(backend, monkeypatch)
    pk = o.reproduce_proving_key(setup, circuit)
    vk = o.reproduce_verification_key(setup, circuit)
    proof = o.reproduce_proof(setup, pk, vk, circuit)
    assert proof["output"] == o.fr_to_string(circuit.output)
    T = o.interpolate_polynomial(o.construct_Omega(d), circuit.trace)
    want = o.g1_mul(o.G1_GENERATOR, o.poly_eval(T, tau))
    assert proof["com_T"] == o.g1_hex(want)
    assert calls["open"] == 15


# ---------------------------------------------------------------------------------------------
# The host-side prover on limb arrays (plonk_amd/prover.py): no restatement code in the loop.
# ---------------------------------------------------------------------------------------------
def test_limb_array_prover_reproduces_the_golden_files(backend, golden):
    from plonk_amd import prover as pv
    setup = o.load_setup_json(golden["srs"])
    backend.srs_load_projective(c.g1_to_limbs(setup.gp.tau_powers_g1))
    circuit = pv.example_circuit()
    p = pv.Prover(backend, 12)
    keys = p.keys(circuit)
    assert [str(x) for x in pv.from_mont(keys["S"])] == golden["proving_key"]["S"][:len(keys["S"])]
    assert [int(x) for x in golden["proving_key"]["S"][len(keys["S"]):]] == [0] * (3 - len(keys["S"]))
    assert [str(x) for x in pv.from_mont(keys["W"])] == golden["proving_key"]["W"]
    assert p.hex_of(keys["com_S"]) == golden["verification_key"]["com_S"]
    assert p.hex_of(keys["com_W"]) == golden["verification_key"]["com_W"]
    assert p.prove(circuit, keys) == golden["proof"]
    assert p.prove_resident(circuit, keys) == golden["proof"]  # polynomials kept on the device


@pytest.mark.parametrize("log_gates", [3, 5])
def test_limb_array_prover_matches_the_restatement(backend, log_gates):
    from plonk_amd import prover as pv
    oc = o.synthetic_circuit(log_gates, 21)
    d = len(oc.trace)
    tau = 0xFEDCBA9876543210 + log_gates
    circuit = pv.Circuit(oc.gates, pv.to_mont(oc.trace), oc.wiring, oc.pub_inputs, oc.output)
    same = pv.synthetic_circuit(log_gates, 21)
    assert np.array_equal(same.trace, circuit.trace) and np.array_equal(same.wiring, circuit.wiring)
    p = pv.Prover(backend, d)
    p.setup(tau)
    keys = p.keys(circuit)
    p.kzg_log = []
    proof = p.prove(circuit, keys)
    # every commitment and opening against the trapdoor, and the verifier's field equations
    assert p.audit_with_tau(tau) == 6 + 14  # T is opened at w rp once for f and g
    shape = (circuit.number_gates, circuit.pub_inputs)
    assert p.check_identities(shape, proof, keys)
    # ... and every opening as the reference's verifier checks it: by pairings, with the public tau_g2 only
    import plonk_amd
    tau_g2 = plonk_amd.kzg_setup_tau_g2(pv.to_mont([tau])[0])
    assert p.verify_openings(tau_g2)
    assert not p.verify_openings(plonk_amd.kzg_setup_tau_g2(pv.to_mont([tau + 1])[0]))
    forged = dict(proof)
    forged["proof_T_S_zero"] = list(proof["proof_T_S_zero"])
    forged["proof_T_S_zero"][1] = str((int(proof["proof_T_S_zero"][1]) + 1) % pv.R_MOD)
    assert not p.check_identities(shape, forged, keys)
    setup = o.Setup(len(oc.gates), 2, 1, d, o.kzg_setup_with_tau(2 * d - 3, tau))
    pk, vk = o.reproduce_proving_key(setup, oc), o.reproduce_verification_key(setup, oc)
    assert [o.fr_to_string(x) for x in pv.from_mont(keys["S"])] == pk["S"]
    assert [o.fr_to_string(x) for x in pv.from_mont(keys["W"])] == pk["W"]
    assert {"com_S": p.hex_of(keys["com_S"]), "com_W": p.hex_of(keys["com_W"])} == vk
    assert proof == o.reproduce_proof(setup, pk, vk, oc)
    # the device-resident pipeline: the same proof, the same audit
    p.kzg_log = []
    assert p.prove_resident(circuit, keys) == proof
    assert p.audit_with_tau(tau) == 20


@pytest.mark.parametrize("log_gates,number_gates", [(3, 4), (3, 5), (4, 9), (5, 30)])
def test_circuits_whose_gate_rows_are_a_proper_prefix_of_their_subgroup(backend, log_gates, number_gates):
    """Any number of gates on d = 3 * 2^k rows: the gate rows Omega[3l], l < number_gates, are then neither the
    subgroup generated by w^3 nor one short of it, so the selector comes from the interpolation on a geometric
    progression and the gate quotient from the division by the product of the root factors.  Keys and proof
    equal the restatement's (plonk/src/setup_proving_key/mod.rs:32-103, plonk/src/prover/part3.rs:32-66), on
    both paths of the prover."""
    from plonk_amd import prover as pv
    oc = o.synthetic_circuit(log_gates, 31, number_gates=number_gates)
    d = len(oc.trace)
    tau = 0x13579BDF02468ACE + number_gates
    circuit = pv.synthetic_circuit(log_gates, 31, number_gates=number_gates)
    assert np.array_equal(circuit.trace, pv.to_mont(oc.trace)) and list(circuit.wiring) == list(oc.wiring)
    p = pv.Prover(backend, d)
    p.setup(tau)
    keys = p.keys(circuit)
    setup = o.Setup(number_gates, 2, 1, d, o.kzg_setup_with_tau(2 * d - 3, tau))
    pk, vk = o.reproduce_proving_key(setup, oc), o.reproduce_verification_key(setup, oc)
    assert [o.fr_to_string(x) for x in pv.from_mont(keys["S"])] == pk["S"][:len(keys["S"])]
    assert all(int(x) == 0 for x in pk["S"][len(keys["S"]):])
    assert [o.fr_to_string(x) for x in pv.from_mont(keys["W"])] == pk["W"]
    assert {"com_S": p.hex_of(keys["com_S"]), "com_W": p.hex_of(keys["com_W"])} == vk
    want = o.reproduce_proof(setup, pk, vk, oc)
    assert p.prove(circuit, keys) == want
    assert p.prove_resident(circuit, keys) == want


def test_a_hundred_thousand_gates_on_rows_for_131071(backend):
    """Beyond the 2^16 points of the quadratic interpolation: 100 000 gates on d = 3 * 2^17 rows.  No restatement
    finishes at this size; the proof is checked as in the full-size test: every commitment and opening against
    the trapdoor, the verifier's field equations, and the openings by pairings with the public parameters."""
    import plonk_amd
    from plonk_amd import prover as pv
    circuit = pv.synthetic_circuit(17, 5, number_gates=100000)
    p = pv.Prover(backend, circuit.d)
    tau = 0x1F2E3D4C5B6A7988
    p.setup(tau)
    keys = p.keys(circuit)
    assert keys["S"].shape[0] <= 100000
    # S takes the gate kinds on the gate rows (evaluated over the subgroup generated by w^3, zero padded)
    n_sub = circuit.d // 3
    padded = np.zeros((n_sub, 4), dtype=np.uint64)
    padded[:keys["S"].shape[0]] = keys["S"]
    on_rows = backend.fr_ntt(padded)[:100000]
    assert np.array_equal(on_rows[circuit.gates != 0], np.tile(p.one, (int((circuit.gates != 0).sum()), 1)))
    assert not on_rows[circuit.gates == 0].any()
    p.kzg_log = []
    proof = p.prove_resident(circuit, keys)
    assert p.audit_with_tau(tau) == 20
    assert p.check_identities((circuit.number_gates, circuit.pub_inputs), proof, keys)
    assert p.verify_openings(plonk_amd.kzg_setup_tau_g2(pv.to_mont([tau])[0]))


def test_limb_array_prover_rejects_what_the_reference_asserts(backend):
    """A wrong gate output, a broken wiring and a wrong claimed output stop the prover where the reference's
    assertions do (prover/part3.rs:61-64, part4.rs:36-42, part5.rs:23-26), on both paths."""
    from plonk_amd import prover as pv
    good = pv.synthetic_circuit(4, 9)
    p = pv.Prover(backend, good.d)
    p.setup(12345)
    keys = p.keys(good)

    def variant(**kw):
        f = dict(gates=good.gates, trace=good.trace.copy(), wiring=good.wiring.copy(), pub_inputs=good.pub_inputs,
                 output=good.output)
        f.update(kw)
        return pv.Circuit(f["gates"], f["trace"], f["wiring"], f["pub_inputs"], f["output"])

    bad_gate = variant()
    # a gate output used nowhere else: only the gate constraint breaks
    last_out = 3 * good.number_gates - 1
    bad_gate.trace[last_out] = pv.to_mont([pv.from_mont(good.trace[last_out])[0] + 1])[0]
    bad_gate.output = (good.output + 1) % pv.R_MOD
    bad_wire = variant()
    j = int(np.flatnonzero(good.wiring != np.arange(good.d))[0])
    bad_wire.trace[j] = pv.to_mont([7])[0]
    bad_output = variant(output=(good.output + 1) % pv.R_MOD)
    for prove in (p.prove, p.prove_resident):
        with pytest.raises(ValueError, match="S_T should cancel"):
            prove(bad_gate, keys)
        with pytest.raises(ValueError, match="T_W should be equal"):
            prove(bad_wire, keys)
        with pytest.raises(ValueError, match="does not match with output"):
            prove(bad_output, keys)
        assert prove(good, keys)["output"] == str(good.output)
