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
