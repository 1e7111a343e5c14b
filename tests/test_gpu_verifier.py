"""`verifier::run` (plonk/src/verifier/mod.rs:13-42) through the library: the reference's own proof
(plonk/data/proof.json) verifies against its verification key and SRS -- field equations plus the 15 pairing
checks folded into one -- tampered proofs do not, and the proofs this build makes verify as well."""
import copy

import numpy as np
import pytest

import plonk_amd
from plonk_amd import verifier
from tests.gpu_util import backend  # noqa: F401

pytestmark = pytest.mark.gpu


def _load_golden_srs(backend, golden):
    backend.srs_load_compressed([bytes.fromhex(h) for h in golden["srs"]["gp"]["tau_powers_g1"]])


def test_the_reference_proof_verifies(backend, golden):
    _load_golden_srs(backend, golden)
    assert verifier.run(backend, golden["srs"], golden["verification_key"], golden["proof"])
    v = verifier.Verifier(backend, 12, 3, 2, bytes.fromhex(golden["srs"]["gp"]["tau_g2"]))
    ok, tuples = v.checks(golden["verification_key"], golden["proof"])
    assert ok and len(tuples) == 15


def test_tampered_proofs_are_rejected(backend, golden):
    _load_golden_srs(backend, golden)
    good, vk, srs = golden["proof"], golden["verification_key"], golden["srs"]
    r = plonk_amd.prover.R_MOD

    def bumped(path, idx=None):
        p = copy.deepcopy(good)
        if idx is None:
            p[path] = str((int(p[path]) + 1) % r)
        elif isinstance(p[path][idx], str) and p[path][idx].isdigit():
            p[path][idx] = str((int(p[path][idx]) + 1) % r)
        return p

    # an opened value in each part, the output, a public input
    for path, idx in (("proof_T_minus_v_zero", 1), ("proof_T_minus_v_zero", 3), ("proof_T_S_zero", 7),
                      ("proof_T_S_zero", 9), ("proof_T_W_prescribed_permutation", 2),
                      ("proof_T_W_prescribed_permutation", 8), ("proof_T_W_prescribed_permutation", 14)):
        assert not verifier.run(backend, srs, vk, bumped(path, idx)), (path, idx)
    assert not verifier.run(backend, srs, vk, bumped("output"))
    p = copy.deepcopy(good)
    p["pub_inputs"][0] = str(int(p["pub_inputs"][0]) + 1)
    assert not verifier.run(backend, srs, vk, p)
    # group elements: two opening proofs swapped (both valid points: only the pairing check can tell), a
    # commitment replaced by another one, an encoding that is no point at all
    p = copy.deepcopy(good)
    a = p["proof_T_S_zero"]
    a[2], a[4] = a[4], a[2]
    assert not verifier.run(backend, srs, vk, p)
    p = copy.deepcopy(good)
    p["proof_T_W_prescribed_permutation"][0] = good["com_T"]
    assert not verifier.run(backend, srs, vk, p)
    p = copy.deepcopy(good)
    p["proof_last_gate_KZG"] = "ff" * 48
    assert not verifier.run(backend, srs, vk, p)
    # the other key
    assert not verifier.run(backend, srs, {"com_S": vk["com_W"], "com_W": vk["com_S"]}, good)
    # the other tau_g2
    other = dict(srs, gp=dict(srs["gp"], tau_g2=plonk_amd.kzg_setup_tau_g2(np.array([5, 0, 0, 0], dtype=np.uint64)).hex()))
    assert not verifier.run(backend, other, vk, good)


@pytest.mark.parametrize("log_gates", [3, 5])
def test_proofs_of_this_build_verify(backend, log_gates):
    from plonk_amd import prover as pv
    circuit = pv.synthetic_circuit(log_gates, 33)
    tau = 0x1F2E3D4C5B6A7988 + log_gates
    p = pv.Prover(backend, circuit.d)
    p.setup(tau)
    keys = p.keys(circuit)
    proof = p.prove_resident(circuit, keys)
    vk = {"com_S": p.hex_of(keys["com_S"]), "com_W": p.hex_of(keys["com_W"])}
    v = verifier.Verifier(backend, circuit.d, circuit.number_gates, len(circuit.pub_inputs),
                          plonk_amd.kzg_setup_tau_g2(pv.to_mont([tau])[0]))
    assert v.run(vk, proof)
    forged = copy.deepcopy(proof)
    forged["output"] = str((int(proof["output"]) + 1) % pv.R_MOD)
    assert not v.run(vk, forged)
