"""GPU parity: the G1 MSM behind kzg_commit / kzg_evaluate and kzg_setup, through the C ABI,
against the oracle and the reference's golden vectors.  Bit-exact (compressed encodings)."""
import random

import numpy as np
import pytest

import plonk_amd
from oracle import capi as c
from oracle import pyref as o
from tests.gpu_util import backend, compress, rand_fr, same_point  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def golden_srs(golden):
    setup = o.load_setup_json(golden["srs"])
    return c.g1_to_limbs(setup.gp.tau_powers_g1)


@pytest.mark.parametrize("window_bits", [0, 2, 3, 8, 13, 15, 16, 17])
def test_verification_key_commitments(backend, golden, golden_srs, window_bits):
    # com_S (3 coefficients) and com_W (12) of plonk/data/verification_key.json
    backend.set_window_bits(window_bits)
    backend.srs_load_projective(golden_srs)
    assert backend.srs_size() == 22
    for name in "SW":
        sc = c.fr_to_mont([int(x) for x in golden["proving_key"][name]])
        assert compress(backend.kzg_commit(sc)).hex() == golden["verification_key"]["com_" + name]
    backend.set_window_bits(0)


def test_srs_round_trip(backend, golden_srs):
    backend.srs_load_projective(golden_srs)
    back = backend.srs_read(0, 22)
    assert c.g1_compress(back) == c.g1_compress(golden_srs)


def test_commit_errors_match_reference(backend, golden_srs):
    backend.srs_load_projective(golden_srs)
    rnd = random.Random(1)
    with pytest.raises(plonk_amd.CommitError):  # plonk/tests/kzg_tests.rs:71-93
        backend.kzg_commit(c.fr_to_mont(rand_fr(rnd, 23)))
    with pytest.raises(plonk_amd.ReferencePanic):  # empty reduce unwrap, kzg.rs:57-58
        backend.kzg_commit(np.zeros((0, 4), dtype=np.uint64))
    with pytest.raises(plonk_amd.ReferencePanic):  # constant polynomial, kzg.rs:85
        backend.kzg_evaluate(c.fr_to_mont([5]), c.fr_to_mont([3])[0])
    fresh = plonk_amd.Backend(device=0)
    with pytest.raises(plonk_amd.PlonkHipError):
        fresh.kzg_commit(c.fr_to_mont([1, 2]))
    fresh.close()


def test_kzg_setup_matches_sequential_chain(backend):
    rnd = random.Random(2)
    tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
    degree = 200
    got = backend.kzg_setup(tau, degree)
    want = c.kzg_setup(tau, degree)  # plonk/src/common/kzg.rs:29-36
    assert c.g1_compress(got) == c.g1_compress(want)
    assert backend.srs_size() == degree + 1
    assert c.g1_compress(backend.srs_read(0, degree + 1)) == c.g1_compress(want)


@pytest.mark.parametrize("n,window_bits", [(1, 0), (2, 4), (257, 0), (1000, 7), (1000, 11), (4096, 0)])
def test_commit_random_polynomials(backend, n, window_bits):
    rnd = random.Random(100 + n + window_bits)
    tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
    backend.set_window_bits(window_bits)
    srs = backend.kzg_setup(tau, n - 1)
    backend.set_window_bits(0)
    sc = c.fr_to_mont(rand_fr(rnd, n))
    got = backend.kzg_commit(sc)
    # plonk/tests/kzg_tests.rs:36-69: commit == G * f(tau)
    assert same_point(got, c.g1_mul(c.g1_generator(), c.fr_eval(sc, tau)))
    if n <= 1000:
        assert same_point(got, c.msm_naive(srs, sc, threads=4))
    # shorter polynomial against the same SRS
    if n > 2:
        m = n // 2 + 1
        assert same_point(backend.kzg_commit(sc[:m]), c.g1_mul(c.g1_generator(), c.fr_eval(sc[:m], tau)))


def test_commit_edge_scalars_and_points(backend):
    rnd = random.Random(3)
    n = 300
    g = c.g1_generator()
    tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
    srs = c.kzg_setup(tau, n - 1)
    # identity points and repeated points inside the SRS
    srs[7] = c.g1_identity()
    srs[8] = srs[9]
    srs[20] = c.g1_neg(srs[21])
    backend.srs_load_projective(srs)
    cases = {
        "zeros": [0] * n,
        "ones": [1] * n,
        "minus_one": [o.R - 1] * n,
        "equal": [rnd.randrange(o.R)] * n,
        "sparse": [rnd.randrange(o.R) if i % 16 else 0 for i in range(n)],
        "small": [rnd.randrange(4) for _ in range(n)],
        "top_heavy": [o.R - 1 - rnd.randrange(1000) for _ in range(n)],
    }
    for name, vals in cases.items():
        sc = c.fr_to_mont(vals)
        assert same_point(backend.kzg_commit(sc), c.msm_naive(srs, sc, threads=4)), name
    assert compress(backend.kzg_commit(c.fr_to_mont(cases["zeros"]))) == o.g1_compress(o.G1_IDENTITY)


@pytest.mark.parametrize("log_n", [12, 16])
def test_commit_skewed_buckets_known_answer(backend, log_n):
    # every scalar equal (all points of a window land in one bucket), all r - 1, one in sixteen zero:
    # commit(f) == [f(tau)] G  (plonk/tests/kzg_tests.rs:36-69)
    rnd = random.Random(6 + log_n)
    n = 1 << log_n
    g = c.g1_generator()
    tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
    backend.kzg_setup(tau, n - 1, want_points=False)
    s = rnd.randrange(o.R)
    cases = {
        "equal": [s] * n,
        "minus_one": [o.R - 1] * n,
        "sparse": [rnd.randrange(o.R) if i % 16 else 0 for i in range(n)],
        "two_values": [s if i % 3 else 5 for i in range(n)],
    }
    for name, vals in cases.items():
        sc = c.fr_to_mont(vals)
        assert same_point(backend.kzg_commit(sc), c.g1_mul(g, c.fr_eval(sc, tau))), name


def test_commit_2_20_points_on_skewed_scalar_sets(backend):
    """The scalar sets that skew the bucket method, at the size of the headline metric: every scalar equal (each
    window's 2^20 digits in ONE bucket, which the fix-up sums by chunks and a join), every scalar r - 1, one in
    sixteen zero, only 0 and 1, all zero, two values in stretches: commit(f) == [f(tau)]G
    (plonk/tests/kzg_tests.rs:36-69) with f(tau) from the oracle."""
    import bench
    n = 1 << 20
    rnd = random.Random(2020)
    g = c.g1_generator()
    tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
    backend.kzg_setup(tau, n - 1, want_points=False)
    sets = dict(bench.adversarial_scalar_sets(n, 97))
    two = np.tile(c.fr_to_mont([rnd.randrange(o.R)])[0], (n, 1))
    two[(np.arange(n) // 5000) % 2 == 1] = c.fr_to_mont([rnd.randrange(o.R)])[0]
    sets["two_values_in_stretches"] = two
    for name, sc in sets.items():
        got = backend.kzg_commit(np.ascontiguousarray(sc))
        assert same_point(got, c.g1_mul(g, c.fr_eval(sc, tau))), name
    # the same sets side by side in one batch (deferred tails: one fix-up, one join, one reduction for all)
    names = list(sets)
    both = backend.kzg_commit_batch([np.ascontiguousarray(sets[k]) for k in names])
    for k, got in zip(names, both):
        assert same_point(got, c.g1_mul(g, c.fr_eval(sets[k], tau))), k


def test_sort_reports_keys_out_of_order(backend):
    """The accumulation checks that the sorted keys it walks are ascending and reports through the context's
    status word if not.  Nothing in the library produces such a list, so the check is exercised through the
    test hook that hands the accumulation an unsorted one: the next wait returns PLONK_HIP_ERR_INTERNAL, the
    context falls back to ballot ranks and keeps working."""
    rnd = random.Random(11)
    n = 4096
    tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
    be = plonk_amd.Backend(device=0)
    try:
        be.kzg_setup(tau, n - 1, want_points=False)
        sc = c.fr_to_mont(rand_fr(rnd, n))
        want = c.g1_mul(c.g1_generator(), c.fr_eval(sc, tau))
        assert same_point(be.kzg_commit(sc), want)
        be.debug_skip_sort(True)
        with pytest.raises(plonk_amd.PlonkHipError) as err:
            be.kzg_commit(sc)
        assert err.value.code == -10
        be.debug_skip_sort(False)
        assert be.sort_rank_mode() == "ballot"
        assert same_point(be.kzg_commit(sc), want)
    finally:
        be.close()


@pytest.mark.parametrize("window_bits", [4, 9, 13, 15, 16, 17])
def test_cooperative_sums_meet_identity_equal_and_opposite_operands(backend, window_bits):
    """The tail sums of small bucket arrays run on four cooperating waves whose barrier loops depend on every
    wave taking the same per-slot decisions.  With every SRS point equal to G (tau = 1) and hand-picked
    scalars the bucket sums that those waves add are equal (doubling path), opposite (sum is the identity),
    or the identity itself, at window widths from a handful of buckets to 2^15 of them (with 16-bit windows
    and 3 or 12 points: the shape of test_verification_key_commitments[16])."""
    g = c.g1_generator()
    backend.set_window_bits(window_bits)
    try:
        for n in (3, 12, 64):
            backend.kzg_setup(c.fr_to_mont([1])[0], n - 1, want_points=False)
            cases = {
                "equal_buckets": [1 + (i & 1) for i in range(n)],            # B_0 = B_1 when n is even
                "opposite": [1 if i & 1 else o.R - 1 for i in range(n)],     # +G and -G meet in bucket 0
                "identity_heavy": [0] * (n - 1) + [5],
                "mixed": [(1, o.R - 1, 2, o.R - 2, 3, 3)[i % 6] for i in range(n)],
                "all_equal": [7] * n,
                "zeros": [0] * n,
            }
            for name, vals in cases.items():
                want = c.g1_mul(g, c.fr_to_mont([sum(vals) % o.R])[0])
                got = backend.kzg_commit(c.fr_to_mont(vals))
                assert same_point(got, want), (name, n, window_bits)
                # the same twice in one batch: the deferred tails run side by side
                both = backend.kzg_commit_batch([c.fr_to_mont(vals), c.fr_to_mont(vals[::-1])])
                assert same_point(both[0], want) and same_point(both[1], want), (name, n, window_bits)
    finally:
        backend.set_window_bits(0)


def test_commit_when_every_srs_point_is_the_generator(backend):
    # tau = 1: all points equal, so additions inside a bucket keep meeting equal operands (the
    # doubling path of the accumulation and of the tree sums); commit(f) == [sum f_i] G
    rnd = random.Random(8)
    n = 1 << 12
    g = c.g1_generator()
    backend.kzg_setup(c.fr_to_mont([1])[0], n - 1, want_points=False)
    for vals in ([1] * n, [rnd.randrange(o.R)] * n, rand_fr(rnd, n), [rnd.randrange(8) for _ in range(n)]):
        sc = c.fr_to_mont(vals)
        total = c.fr_to_mont([sum(vals) % o.R])[0]
        assert same_point(backend.kzg_commit(sc), c.g1_mul(g, total))


def test_open_matches_reference_definition(backend):
    rnd = random.Random(4)
    for n in (2, 3, 65, 300, 5000):
        tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
        backend.kzg_setup(tau, n - 1, want_points=False)
        f = rand_fr(rnd, n)
        u = rnd.randrange(o.R)
        v, proof = backend.kzg_evaluate(c.fr_to_mont(f), c.fr_to_mont([u])[0])
        assert c.fr_from_mont(v)[0] == o.poly_eval(f, u)  # kzg_tests.rs:140
        q, _ = o.poly_divmod(o.poly_sub(f, [o.poly_eval(f, u)]), [(-u) % o.R, 1])
        q_at_tau = c.fr_eval(c.fr_to_mont(q), tau)
        assert same_point(proof, c.g1_mul(c.g1_generator(), q_at_tau))  # kzg_tests.rs:141-145
    # several openings of one polynomial
    us = rand_fr(rnd, 3)
    vs, proofs = backend.kzg_evaluate_many(c.fr_to_mont(f), c.fr_to_mont(us))
    for i, uu in enumerate(us):
        v1, p1 = backend.kzg_evaluate(c.fr_to_mont(f), c.fr_to_mont([uu])[0])
        assert np.array_equal(v1, vs[i]) and same_point(p1, proofs[i])


def test_commit_2_16_points(backend):
    # a single KZG commit: 2^16-point G1 MSM against a synthetic SRS
    rnd = random.Random(5)
    n = 1 << 16
    tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
    srs = backend.kzg_setup(tau, n - 1)
    # spot-check the generated SRS against independent scalar multiplications
    g = c.g1_generator()
    for i in (0, 1, 2, 12345, n - 1):
        assert same_point(srs[i], c.g1_mul(g, c.fr_to_mont([pow(c.fr_from_mont(tau)[0], i, o.R)])[0]))
    sc = np.frombuffer(rnd.randbytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)  # any value below r is a valid Montgomery residue
    got = backend.kzg_commit(sc)
    assert same_point(got, c.g1_mul(g, c.fr_eval(sc, tau)))
    aff = c.g1_to_affine(srs)
    assert same_point(got, c.msm_pippenger(aff, sc, c=13, threads=8))
    # device-resident entry point
    d_sc = backend.to_device(sc)
    d_out = backend.malloc(144)
    backend.kzg_commit_dev(d_sc, n, d_out)
    backend.sync()
    assert same_point(backend.from_device(d_out, (18,)), got)
    backend.free(d_sc)
    backend.free(d_out)


@pytest.mark.parametrize("window_bits", [0, 9])
def test_commit_at_every_kind_of_segment_length(backend, window_bits):
    """The accumulation cuts the sorted (bucket, entry) list into segments of one length per commitment, which
    the library fits to the lanes the device keeps resident (any value from 1 up, odd ones included).  The
    measurement override walks lengths the fit would not pick at this size: one position per lane, odd and even
    short ones, powers of two, one longer than the whole list of a window.  commit(f) == [f(tau)]G
    (plonk/tests/kzg_tests.rs:36-69) for uniform scalars and for a set with two long buckets per window."""
    rnd = random.Random(31 + window_bits)
    n = 1 << 16
    g = c.g1_generator()
    tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
    backend.set_window_bits(window_bits)
    backend.kzg_setup(tau, n - 1, want_points=False)
    uniform = np.frombuffer(rnd.randbytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    uniform[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    two = np.tile(c.fr_to_mont([rnd.randrange(o.R)])[0], (n, 1))
    two[(np.arange(n) // 777) % 2 == 1] = c.fr_to_mont([rnd.randrange(o.R)])[0]
    want = {"uniform": c.g1_mul(g, c.fr_eval(uniform, tau)), "two": c.g1_mul(g, c.fr_eval(two, tau))}
    try:
        for seg_len in (1, 2, 3, 7, 16, 30, 31, 32, 33, 255, 256, 1000, 70001):
            # the library's way of fetching the pairs for this length, then both ways forced (one position at a
            # time; groups of eight through LDS, whose 16-byte alignment differs from lane to lane for lengths
            # that are no multiple of four)
            for pair_loads in (0, 1, 2):
                backend.bench_msm_tuning(seg_len, pair_loads)
                assert same_point(backend.kzg_commit(uniform), want["uniform"]), (seg_len, pair_loads)
                assert same_point(backend.kzg_commit(two), want["two"]), (seg_len, pair_loads)
            both = backend.kzg_commit_batch([uniform, two])
            assert same_point(both[0], want["uniform"]) and same_point(both[1], want["two"]), seg_len
    finally:
        backend.bench_msm_tuning(0, 0)
        backend.set_window_bits(0)


def test_commit_2_20_points_known_answer(backend):
    rnd = random.Random(6)
    n = 1 << 20
    tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
    backend.kzg_setup(tau, n - 1, want_points=False)
    sc = np.frombuffer(rnd.randbytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    got = backend.kzg_commit(sc)
    assert same_point(got, c.g1_mul(c.g1_generator(), c.fr_eval(sc, tau)))


def test_commit_2_24_points_known_answer(backend):
    # the 2^24-point G1 MSM of the benchmark: size-independent known answer commit == [f(tau)]G
    n = 1 << 24
    tau = c.fr_to_mont([0x1234567 + (1 << 200)])[0]
    backend.kzg_setup(tau, n - 1, want_points=False)
    rng = np.random.default_rng(24)
    sc = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    got = backend.kzg_commit(sc)
    assert same_point(got, c.g1_mul(c.g1_generator(), c.fr_eval(sc, tau)))
    # two half-range shards (SRS slices generated independently) sum to the same point
    half = n // 2
    d_out = backend.malloc(2 * 144)
    parts = []
    for r in range(2):
        backend.kzg_setup_range(tau, r * half, half)
        parts.append(backend.kzg_commit(sc[r * half:(r + 1) * half]))
    d_parts = backend.to_device(np.stack(parts))
    backend.g1_sum_dev(d_parts, 2, d_out)
    backend.sync()
    assert same_point(backend.from_device(d_out, (18,)), got)
    backend.free(d_parts)
    backend.free(d_out)


# (13, 2, 20): slices of 2^18 buckets, which the reduction sums in two levels
@pytest.mark.parametrize("log_n,ranks,window_bits", [(18, 4, 0), (14, 2, 9), (12, 8, 6), (13, 2, 20)])
def test_bucket_partial_exchange(backend, log_n, ranks, window_bits):
    """The multi-GPU exchange of bucket partials, with the ranks played one after the other on this
    GPU: every rank stops after its bucket sums, slice g of every rank's array goes to rank g, the
    slice shares add up to the commitment."""
    n = 1 << log_n
    per = n // ranks
    rnd = random.Random(40 + log_n)
    tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
    sc = np.frombuffer(rnd.randbytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    sc[5] = 0
    sc[per:per + 40] = sc[7]  # a crowded bucket on rank 1
    backend.set_window_bits(window_bits)
    arrays = []
    for r in range(ranks):
        backend.kzg_setup_range(tau, r * per, per)
        nb, each = backend.msm_bucket_layout()
        assert nb == backend.msm_plan()["buckets"]
        d_sc = backend.to_device(sc[r * per:(r + 1) * per])
        d_b = backend.malloc(nb * each)
        backend.msm_buckets_dev(d_sc, per, d_b)
        backend.sync()
        arrays.append(backend.from_device(d_b, (nb, each), dtype=np.uint8))
        backend.free(d_sc)
        backend.free(d_b)
    count = nb // ranks
    d_shares = backend.malloc(ranks * 144)
    for g in range(ranks):  # what rank g holds after the all_to_all
        parts = np.stack([a[g * count:(g + 1) * count] for a in arrays])
        d_parts = backend.to_device(parts)
        backend.msm_reduce_bucket_slice_dev(d_parts, ranks, g * count, count, d_shares + g * 144)
        backend.sync()
        backend.free(d_parts)
    d_out = backend.malloc(144)
    backend.g1_sum_dev(d_shares, ranks, d_out)
    backend.sync()
    got = backend.from_device(d_out, (18,))
    assert same_point(got, c.g1_mul(c.g1_generator(), c.fr_eval(sc, tau)))
    # one part, whole array: the same as the single-GPU commitment of that rank's range
    d_parts = backend.to_device(arrays[-1])
    backend.msm_reduce_bucket_slice_dev(d_parts, 1, 0, nb, d_out)
    backend.sync()
    assert same_point(backend.from_device(d_out, (18,)), backend.kzg_commit(sc[(ranks - 1) * per:]))
    with pytest.raises(plonk_amd.PlonkHipError):
        backend.msm_reduce_bucket_slice_dev(d_parts, 1, count // 2, count, d_out)  # misaligned slice
    for ptr in (d_parts, d_shares, d_out):
        backend.free(ptr)
    backend.set_window_bits(0)


def test_commit_2_26_points_known_answer(backend):
    # the largest size whose 12 window tables (96 GiB) fit beside the work buffers in 288 GB
    n = 1 << 26
    tau = c.fr_to_mont([0x7654321 + (1 << 190)])[0]
    backend.kzg_setup(tau, n - 1, want_points=False)
    rng = np.random.default_rng(26)
    sc = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    got = backend.kzg_commit(sc)
    assert same_point(got, c.g1_mul(c.g1_generator(), c.fr_eval(sc, tau)))


def test_compiled_host_program_on_the_cpp_wrapper(backend):
    """examples/kzg_roundtrip.cpp (built by the Makefile with plain g++ against include/plonk_hip.hpp):
    setup, commit, open, error behaviour from a compiled caller, checked against the oracle."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(plonk_amd.__file__), "_lib", "kzg_roundtrip")
    assert os.path.exists(exe), "run make -C plonk_amd/csrc"
    rnd = random.Random(77)
    n = 700
    tau_i, u_i = rnd.randrange(1, o.R), rnd.randrange(o.R)
    tau, u = c.fr_to_mont([tau_i])[0], c.fr_to_mont([u_i])[0]
    coeffs = c.fr_to_mont(rand_fr(rnd, n))
    args = [exe, str(n - 1)] + ["%x" % int(x) for x in tau] + ["%x" % int(x) for x in u]
    res = subprocess.run(args, input=coeffs.tobytes(), capture_output=True, timeout=120)
    assert res.returncode == 0, res.stderr.decode()
    lines = res.stdout.decode().split("\n")
    assert lines[0] == "srs %d" % n
    f_tau, v = c.fr_eval(coeffs, tau), c.fr_eval(coeffs, u)
    assert lines[1] == c.g1_compress(c.g1_mul(c.g1_generator(), f_tau).reshape(1, 18))[0].hex()
    assert [int(x, 16) for x in lines[2].split()] == [int(x) for x in v]
    quotient_at_tau = (c.fr_from_mont(f_tau.reshape(1, 4))[0] - c.fr_from_mont(v.reshape(1, 4))[0]) * pow(tau_i - u_i, -1, o.R) % o.R
    want = c.g1_mul(c.g1_generator(), c.fr_to_mont([quotient_at_tau])[0])
    assert lines[3] == c.g1_compress(want.reshape(1, 18))[0].hex()
    assert lines[4] == "roundtrip ok"
    assert lines[5] == "verify ok"
    assert lines[6] == "polynomials ok"
    assert lines[7].startswith("commit_error ") and "MISSING" not in lines[7]
    assert lines[8] == "panic ok"


@pytest.mark.parametrize("n", [(1 << 23) + 12345, (1 << 24) + 777])
def test_large_host_commit_in_slices_equals_the_resident_commit(backend, n):
    """Host coefficients of 2^23 and more go to the device in slices whose copies overlap the previous
    slice's accumulation (each slice an MSM against its SRS range, tails side by side, partials added):
    same commitment as the one-piece path on resident coefficients, and the known answer [f(tau)]G.
    The two sizes: two equal slices; four slices of an eighth, an eighth, a quarter and the rest."""
    rng = np.random.default_rng(23)
    tau = c.fr_to_mont([0xABCDEF + (1 << 190)])[0]
    backend.kzg_setup(tau, n - 1, want_points=False)
    sc = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    got = backend.kzg_commit(sc)
    d = backend.to_device(sc)
    out = backend.malloc(144)
    try:
        backend.kzg_commit_dev(d, n, out)
        backend.sync()
        assert same_point(got, backend.from_device(out, (18,)))
    finally:
        backend.free(d)
        backend.free(out)
    assert same_point(got, c.g1_mul(c.g1_generator(), c.fr_eval(sc, tau)))


def test_large_host_opening_in_slices(backend):
    """kzg_evaluate on 2^24 and more host coefficients copies them in four slices from the top down, scans
    each with the value carried from the slice above and commits the quotient slice by slice: v = f(u) and
    the proof [(f(tau) - v) / (tau - u)]G by the oracle's field arithmetic, and the same proof as the
    one-piece opening of the resident polynomial."""
    n = (1 << 24) + 777
    rng = np.random.default_rng(29)
    tau_int, u_int = 0xFEDCBA + (1 << 180), 0x13579B + (1 << 222)
    tau, u = c.fr_to_mont([tau_int])[0], c.fr_to_mont([u_int])[0]
    backend.kzg_setup(tau, n - 2, want_points=False)  # the quotient has n - 1 coefficients
    sc = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    v, proof = backend.kzg_evaluate(sc, u)
    want_v = c.fr_eval(sc, u)
    assert np.array_equal(np.asarray(v).reshape(4), want_v)
    f_tau = c.fr_from_mont(c.fr_eval(sc, tau))[0]
    q_tau = (f_tau - c.fr_from_mont(want_v)[0]) * pow(tau_int - u_int, -1, o.R) % o.R
    assert same_point(proof, c.g1_mul(c.g1_generator(), c.fr_to_mont([q_tau])[0]))
    d = backend.to_device(sc)
    try:
        vs_dev, proofs_dev = backend.kzg_open_many_dev(d, n, np.asarray(u).reshape(1, 4))
        assert np.array_equal(vs_dev[0], want_v)
        assert same_point(proof, proofs_dev[0])
    finally:
        backend.free(d)


def test_full_kzg_protocol_like_the_reference_test(backend):
    """plonk/tests/kzg_tests.rs:199-227: ten random degrees up to 100: setup, commit, evaluate at a random point,
    verify by pairings -- every step through the library.  (A constant polynomial has an empty quotient: the
    reference's kzg_evaluate panics on it, kzg.rs:85.)"""
    rnd = random.Random(199)
    for degree in [1, 2, 100] + [rnd.randrange(1, 101) for _ in range(7)]:
        tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
        backend.kzg_setup(tau, degree, want_points=False)
        f = c.fr_to_mont(rand_fr(rnd, degree + 1))
        com_f = backend.kzg_commit(f)
        u = c.fr_to_mont([rnd.randrange(o.R)])[0]
        v, proof = backend.kzg_evaluate(f, u)
        tau_g2 = plonk_amd.kzg_setup_tau_g2(tau)
        assert plonk_amd.kzg_verify(tau_g2, com_f, u, v, proof), degree
        assert not plonk_amd.kzg_verify(tau_g2, com_f, u, c.fr_to_mont([1])[0], proof), degree
    backend.kzg_setup(c.fr_to_mont([5])[0], 0, want_points=False)
    with pytest.raises(plonk_amd.ReferencePanic):
        backend.kzg_evaluate(c.fr_to_mont([7]), c.fr_to_mont([3])[0])


@pytest.mark.parametrize("window_bits", [0, 3, 6, 7, 8, 9, 13, 15, 16, 17, 20, 22, 23])
def test_first_sort_pass_made_from_the_scalars(window_bits, monkeypatch):
    """Large commitments take the first pass of their bucket sort straight from the scalars; a context created with
    PLONK_HIP_FRONT_END=scalars does so at every size whose shape allows it (a multiple of 2048 scalars).  The
    result must be the point the pairs-then-sort path gives, and [f(tau)]G (plonk/tests/kzg_tests.rs:36-69),
    for uniform, repeated, zero and r - 1 scalars, and for a shorter polynomial; and the same point again when the
    sort ranks its keys by ballots (PLONK_HIP_SORT_RANK=ballot) instead of LDS adds."""
    rnd = random.Random(900 + window_bits)
    n = 3 * 2048
    tau = c.fr_to_mont([rnd.randrange(1, o.R)])[0]
    g = c.g1_generator()
    s = rnd.randrange(o.R)
    cases = {
        "uniform": rand_fr(rnd, n),
        "equal": [s] * n,
        "edge": [(0, 1, o.R - 1, (o.R - 1) // 2, (o.R + 1) // 2, 1 << 254, s)[i % 7] for i in range(n)],
        "sparse": [rnd.randrange(o.R) if i % 16 else 0 for i in range(n)],
    }
    results = {}
    for mode in ("scalars", "pairs", "ballot"):
        # third run: the first pass from the scalars again, keys ranked by ballots instead of LDS adds
        monkeypatch.setenv("PLONK_HIP_FRONT_END", "scalars" if mode == "ballot" else mode)
        monkeypatch.setenv("PLONK_HIP_SORT_RANK", "ballot" if mode == "ballot" else "")
        be = plonk_amd.Backend(device=0)
        assert mode != "ballot" or be.sort_rank_mode() == "ballot"
        try:
            be.set_window_bits(window_bits)
            be.kzg_setup(tau, n + 99, want_points=False)
            for name, vals in cases.items():
                sc = c.fr_to_mont(vals)
                results[mode, name] = compress(be.kzg_commit(sc))
                results[mode, name, "short"] = compress(be.kzg_commit(sc[:2048]))
        finally:
            be.close()
    for name, vals in cases.items():
        sc = c.fr_to_mont(vals)
        for other in ("pairs", "ballot"):
            assert results["scalars", name] == results[other, name], (name, other)
            assert results["scalars", name, "short"] == results[other, name, "short"], (name, other)
        assert results["scalars", name] == compress(c.g1_mul(g, c.fr_eval(sc, tau))), name
