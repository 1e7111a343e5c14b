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


def test_example_proof_bytes_with_the_polynomial_layer_on_the_gpu(backend, golden, monkeypatch):
    """The same proof, now with interpolation, f(wX), products and both quotients of
    common/polynomials.rs and common/protocols.rs also computed on the GPU (the gadget logic,
    the challenges and the JSON stay with the restatement)."""
    setup = o.load_setup_json(golden["srs"])
    backend.srs_load_projective(c.g1_to_limbs(setup.gp.tau_powers_g1))
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
        if list(xs) == o.construct_Omega(n):
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
    assert o.reproduce_proving_key(setup) == golden["proving_key"]
    assert o.reproduce_verification_key(setup) == golden["verification_key"]
    proof = o.reproduce_proof(setup, golden["proving_key"], golden["verification_key"])
    assert proof == golden["proof"]
    assert calls["commit"] == 8 and calls["open"] == 15
    assert min(calls.values()) > 0, calls
