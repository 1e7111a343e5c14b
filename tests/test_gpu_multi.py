"""Several GPUs behind the C ABI in one process (plonk_hip_multi_*): with one sub-context the results are
those of the single-GPU entry points; with two or three sub-contexts (here all on this box's one GPU: the
split, the peer copies of the partials and the chained scan carries are the same) they still are, and
they match the known answers commit(f) = [f(tau)]G (plonk/tests/kzg_tests.rs:36-69) and
proof = [q(tau)]G (:95-146) from the oracle."""
import numpy as np
import pytest

import plonk_amd
from oracle import capi as c
from oracle import pyref as o
from tests.gpu_util import backend, same_point  # noqa: F401

pytestmark = pytest.mark.gpu


def rand(rng, n):
    a = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    return a


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
@pytest.mark.parametrize("n", [7, 1000, 70001])
def test_commit_and_open_equal_the_single_gpu_results(backend, devices, n):
    rng = np.random.default_rng(n + len(devices))
    tau = rand(rng, 1)[0]
    backend.kzg_setup(tau, n - 1, want_points=False)
    m = plonk_amd.MultiBackend(devices)
    try:
        pts = m.kzg_setup(tau, n - 1)
        assert m.device_count() == len(devices)
        spans = [m.srs_range(g) for g in range(len(devices))]
        assert spans[0][0] == 0 and sum(cnt for _, cnt in spans) == n
        assert c.g1_compress(pts) == c.g1_compress(backend.srs_read(0, n))  # representatives may differ
        for length in sorted({n, n - 1, max(2, n // 2), 2, 3}):
            f = rand(rng, length)
            assert same_point(m.kzg_commit(f), backend.kzg_commit(f)), length
            u = rand(rng, 1)[0]
            v, proof = m.kzg_evaluate(f, u)
            want_v, want_proof = backend.kzg_evaluate(f, u)
            assert np.array_equal(v, want_v), length
            assert same_point(proof, want_proof), length
        coms = m.kzg_commit_batch([rand(rng, 5), rand(rng, n)])
        assert coms.shape == (2, 18)
    finally:
        m.close()


def test_known_answers_and_errors_with_two_sub_contexts():
    c.build()
    rng = np.random.default_rng(99)
    n = 4097
    tau_int = 0xC0FFEE + (1 << 200)
    tau = c.fr_to_mont([tau_int])[0]
    m = plonk_amd.MultiBackend([0, 0])
    try:
        m.kzg_setup(tau, n - 1, want_points=False)
        f = rand(rng, n)
        want = c.g1_mul(c.g1_generator(), c.fr_eval(f, tau))
        assert same_point(m.kzg_commit(f), want)
        u = rand(rng, 1)[0]
        v, proof = m.kzg_evaluate(f, u)
        assert np.array_equal(v, c.fr_eval(f, u))
        # q(tau) = (f(tau) - f(u)) / (tau - u)
        fi, ui, vi = c.fr_from_mont(c.fr_eval(f, tau).reshape(1, 4))[0], c.fr_from_mont(u.reshape(1, 4))[0], \
            c.fr_from_mont(v.reshape(1, 4))[0]
        q_tau = (fi - vi) * pow(tau_int - ui, -1, o.R) % o.R
        assert same_point(proof, c.g1_mul(c.g1_generator(), c.fr_to_mont([q_tau])[0]))
        with pytest.raises(plonk_amd.CommitError):  # kzg.rs:49-50
            m.kzg_commit(rand(rng, n + 1))
        with pytest.raises(plonk_amd.ReferencePanic):  # kzg.rs:57-58
            m.kzg_commit(np.zeros((0, 4), dtype=np.uint64))
        with pytest.raises(plonk_amd.ReferencePanic):  # kzg.rs:85
            m.kzg_evaluate(rand(rng, 1), u)
    finally:
        m.close()


def test_srs_loaded_from_projective_points_is_split_by_range(backend):
    rng = np.random.default_rng(3)
    n = 301
    tau = rand(rng, 1)[0]
    pts = backend.kzg_setup(tau, n - 1)
    m = plonk_amd.MultiBackend([0, 0, 0])
    try:
        m.srs_load_projective(pts)
        f = rand(rng, n)
        assert same_point(m.kzg_commit(f), backend.kzg_commit(f))
    finally:
        m.close()


@pytest.mark.parametrize("staged", [False, True])
def test_uneven_ranges_fewer_points_than_devices_and_staged_partials(backend, staged, monkeypatch):
    """Three sub-contexts with ranges of different lengths, an SRS of fewer points than sub-contexts (the last
    hold none), polynomials that end inside the first range (the other devices contribute nothing: zero
    slices), and the same with the partials staged through pinned host memory (the path of a device without
    peer access to the first one, forced by PLONK_HIP_MULTI_STAGED=1)."""
    monkeypatch.setenv("PLONK_HIP_MULTI_STAGED", "1" if staged else "0")
    rng = np.random.default_rng(17)
    for n in (2, 3, 4, 11, 1000):
        tau = rand(rng, 1)[0]
        backend.kzg_setup(tau, n - 1, want_points=False)
        m = plonk_amd.MultiBackend([0, 0, 0])
        try:
            m.kzg_setup(tau, n - 1, want_points=False)
            spans = [m.srs_range(g) for g in range(3)]
            assert sum(cnt for _, cnt in spans) == n and [f for f, _ in spans] == sorted(f for f, _ in spans)
            for length in sorted({n, max(2, n - 1), 2, max(2, spans[0][1]), max(2, spans[0][1] + 1)}):
                if length > n:
                    continue
                f = rand(rng, length)
                assert same_point(m.kzg_commit(f), backend.kzg_commit(f)), (n, length)
                u = rand(rng, 1)[0]
                v, proof = m.kzg_evaluate(f, u)
                want_v, want_proof = backend.kzg_evaluate(f, u)
                assert np.array_equal(v, want_v) and same_point(proof, want_proof), (n, length)
            one = rand(rng, 1)
            assert same_point(m.kzg_commit(one), backend.kzg_commit(one))
        finally:
            m.close()
