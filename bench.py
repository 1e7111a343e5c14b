#!/usr/bin/env python3
"""Headline benchmark: G1 MSM throughput (Mpts/s) of one KZG commitment.

Default workload (--workload msm): one 2^24-point BLS12-381 G1 MSM against a synthetic
powers-of-tau SRS, sharded by point range over the N GPUs of one node.  Rank g keeps SRS points
[g*n/N, (g+1)*n/N) resident (window tables built once, outside the timed region) and the
matching coefficient slice resident in HBM; a step is one full commitment:
    local MSM on every GPU -> gather of the N partial points (RCCL all_gather, 144 B each)
    -> sum of the partials on rank 0.
Total work is fixed as N grows ("strong" scaling).  Prints ONE JSON line on rank 0.

--workload ntt / quotient: the batch of 16 transforms of 2^20 points (or 16 coset quotients
2^21 -> 2^20 coefficients), polynomial b on rank b mod N, no collective on the data path.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 5 --warmup 2
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

# Algorithmic work per MSM point in the canonical accounting: 16 signed 16-bit windows x one mixed
# addition (8M + 2S = 10 Fq multiplications) x 300 32-bit multiply-adds per 12-limb Montgomery
# multiplication.  The plan actually run may use fewer, wider windows (12 at 2^24): the issued work
# is reported next to it.
MADS_PER_FQ_MUL = 300
FQ_MULS_PER_ADDITION = 10
CANONICAL_WINDOWS = 16
MADS_PER_POINT = CANONICAL_WINDOWS * FQ_MULS_PER_ADDITION * MADS_PER_FQ_MUL
BYTES_PER_POINT = 96 + 32  # affine point + scalar
FR_MADS_PER_MUL = 136      # 8 x 8 product + 8 x 8 + 8 reduction on 32-bit limbs
HBM_PEAK_GBS = 8000.0
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic.json")

FR_MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def fr_limbs(value):
    return np.array([(value >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


FR_MINUS_ONE = fr_limbs(FR_MODULUS - (1 << 256) % FR_MODULUS)  # -1 in Montgomery form
FR_ZERO = np.zeros(4, dtype=np.uint64)


def measured_traffic(kernel, workload_key):
    """HBM bytes per launch (or per step) from the counter passes of scripts/profile.sh for this kernel and
    workload, with the commit they were taken at; (None, why) when there is no such record."""
    try:
        with open(TRAFFIC_FILE) as fh:
            rec = json.load(fh)
        entry = rec["entries"][workload_key][kernel]
        return int(entry["bytes"]), "rocprofv3 counter passes (size-resolved TCC_EA0_RDREQ_{128,64,32}B + WRITE_SIZE) of build %s: %s" % (
            rec.get("commit", "?"), entry.get("what", ""))
    except (OSError, KeyError, ValueError):
        return None, "no counter record for this workload in profiles/traffic.json (scripts/profile.sh writes it)"


def hbm_view(be, n_local, windows, acc_ms):
    """Memory side of k_accumulate: one 128-byte table line plus one sorted (bucket, entry) pair per
    point and window, against the HBM peak and against the measured random-line gather rate for a
    table of this size."""
    bytes_per_launch = n_local * windows * (128 + 8)
    achieved = bytes_per_launch / (acc_ms * 1e-3) / 1e9
    # odd size argument: the kernel's own access form (eight lanes per line, global-to-LDS loads)
    gather = be.bench_gather_rate(min(n_local * windows * 128, 48 << 30) | 1) / 1e9
    return {"achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "bytes_per_launch": bytes_per_launch,
            "random_line_gather_ceiling": round(gather, 1), "frac_of_gather_ceiling": round(achieved / gather, 3)}


def synthetic_scalars(n, seed):
    """Uniform 254-bit values used directly as Montgomery residues (all below r)."""
    rng = np.random.default_rng(seed)
    sc = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    sc[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    return sc


FR_ONE = fr_limbs((1 << 256) % FR_MODULUS)  # 1 in Montgomery form


def adversarial_scalar_sets(n, seed):
    """Scalar sets that skew the bucket method (Montgomery residues, n x 4 limbs): every scalar equal
    (each window's digits fall into ONE bucket), every scalar r - 1, one scalar in sixteen zero,
    only the values 0 and 1, and all zero."""
    rng = np.random.default_rng(seed)
    one_value = synthetic_scalars(1, seed + 1)[0]
    sparse = synthetic_scalars(n, seed + 2)
    sparse[::16] = 0
    bits = rng.integers(0, 2, size=n, dtype=np.uint64)
    return {
        "all_equal": np.tile(one_value, (n, 1)),
        "all_r_minus_1": np.tile(FR_MINUS_ONE, (n, 1)),
        "one_in_16_zero": sparse,
        "zeros_and_ones": bits[:, None] * FR_ONE[None, :],
        "all_zero": np.zeros((n, 4), dtype=np.uint64),
    }


def time_adversarial(be, log_n, tau, reps, uniform_ms=None, d_buf=None):
    """One resident commitment per skewed scalar set on a context whose SRS (powers of tau) is loaded:
    milliseconds by HIP events, the ratio to the uniform-scalar time, and the result against the known
    answer [f(tau)]G (plonk/tests/kzg_tests.rs:36-69; f(tau) by the library's evaluation kernel, then a
    one-coefficient commitment)."""
    n = 1 << log_n
    d_sc = d_buf if d_buf else be.malloc(n * 32)
    d_out = be.malloc(144)
    out = {}
    if uniform_ms is None:
        be.to_device(synthetic_scalars(n, 4242), d_sc)
        be.kzg_commit_dev(d_sc, n, d_out)
        be.sync()
        uniform_ms = gpu_time(be, lambda: be.kzg_commit_dev(d_sc, n, d_out), reps)
    out["uniform_ms"] = round(uniform_ms, 3)
    worst = 0.0
    for name, sc in adversarial_scalar_sets(n, 77 + log_n).items():
        be.to_device(sc, d_sc)
        be.kzg_commit_dev(d_sc, n, d_out)
        be.sync()
        got = be.from_device(d_out, (18,))
        want = be.kzg_commit(np.ascontiguousarray(be.fr_eval(sc, tau)).reshape(1, 4))
        enc = be.g1_compress(np.stack([np.asarray(want).reshape(18), got.reshape(18)]))
        ms = gpu_time(be, lambda: be.kzg_commit_dev(d_sc, n, d_out), reps)
        worst = max(worst, ms / uniform_ms)
        out[name] = {"ms": round(ms, 3), "vs_uniform": round(ms / uniform_ms, 3),
                     "equals_f_of_tau_times_G": bool(enc[0] == enc[1])}
    out["worst_vs_uniform"] = round(worst, 3)
    be.free(d_out)
    if not d_buf:
        be.free(d_sc)
    return out


def cpu_baseline(be, n_local, sample_log, threads, log_n_total):
    """The reference's commitment algorithm (one double-and-add per coefficient, then the sum;
    plonk/src/common/kzg.rs:53-58) restated in C (oracle/oracle.c), timed on a bounded sample."""
    from oracle import capi as c

    c.build()
    m = min(1 << sample_log, n_local)
    pts = be.srs_read(0, m)
    sc = synthetic_scalars(m, 99)
    t0 = time.perf_counter()
    c.msm_naive(pts, sc, threads=threads)
    dt = time.perf_counter() - t0
    out = {
        "value": m / dt / 1e6,
        "unit": "Mpts/s",
        "cores": threads,
        "host_cpu_count": os.cpu_count(),
        "kind": "port",
        "sample": "2^%d points of the same SRS, naive per-coefficient double-and-add + sum "
                  "(the reference's kzg_commit loop), %.2f s" % (sample_log, dt),
        # the loop is exactly linear in the number of coefficients
        "extrapolated_s_for_2^%d" % log_n_total: round(dt * (1 << log_n_total) / m, 1),
        "extrapolation": "linear in n from the measured sample (restatement, extrapolated)",
    }
    # the same algorithm class as the GPU path, for a fairer second figure
    m2 = min(1 << (sample_log + 2), n_local)
    aff = c.g1_to_affine(be.srs_read(0, m2))
    sc2 = synthetic_scalars(m2, 98)
    t0 = time.perf_counter()
    c.msm_pippenger(aff, sc2, c=13, threads=threads)
    dt2 = time.perf_counter() - t0
    out["cpu_pippenger"] = {"value": m2 / dt2 / 1e6, "unit": "Mpts/s", "cores": threads,
                            "sample": "2^%d points, signed-digit bucket method c=13, %.2f s" % (
                                m2.bit_length() - 1, dt2)}
    return out


def cpu_baseline_ntt(threads, lagrange_log=12, ntt_log=20):
    """What the reference does where the GPU path runs a transform: Lagrange interpolation
    (plonk/src/common/polynomials.rs:14-58, quadratic) restated in C, timed at a bounded size, and the
    oracle's radix-2 transform at the benchmark's size as the same-algorithm-class figure."""
    from oracle import capi as c

    c.build()
    m = 1 << lagrange_log
    w = int(c.fr_from_mont(c.root_of_unity(lagrange_log).reshape(1, 4))[0])
    xs = [1]
    for _ in range(m - 1):
        xs.append(xs[-1] * w % FR_MODULUS)
    xs = c.fr_to_mont(xs)
    ys = synthetic_scalars(m, 5)
    t0 = time.perf_counter()
    c.interpolate_lagrange(xs, ys, threads=threads)
    dt = time.perf_counter() - t0
    data = synthetic_scalars(1 << ntt_log, 6)
    t0 = time.perf_counter()
    c.fr_ntt(data)
    dt2 = time.perf_counter() - t0
    return {
        "lagrange_interpolation": {
            "value": round(dt * 1e3, 1), "unit": "ms", "cores": threads, "kind": "port",
            "sample": "interpolate_polynomial over the 2^%d-point Omega as the reference computes it" % lagrange_log,
            "extrapolated_ms_for_2^%d" % ntt_log: round(dt * 1e3 * 4.0 ** (ntt_log - lagrange_log), 0),
            "extrapolation": "quadratic in n from the measured sample (restatement, extrapolated)"},
        "radix2_ntt_2^%d" % ntt_log: {"value": round(dt2 * 1e3, 1), "unit": "ms", "cores": 1, "kind": "port",
                                      "sample": "one forward transform, oracle/oracle.c"},
    }


CLOCK_WARM_MS = 80.0


def gpu_time(be, fn, reps, before=None, warm_ms=CLOCK_WARM_MS):
    """Average GPU milliseconds of fn() over reps runs, by HIP events on the library's stream;
    before() (untimed) runs ahead of every repetition.  The same work runs untimed for warm_ms first: the
    shader clock needs about 60 ms of continuous load to come up (and 20 ms of idling to fall back), so a
    short burst after host-side preparation would be timed at two thirds of the clock the kernels hold."""
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < warm_ms:
        if before:
            before()
        fn()
        be.sync()
    total = 0.0
    for _ in range(reps):
        if before:
            before()
        be.timer_start()
        fn()
        total += be.timer_stop()
    return total / reps


def transform_rooflines(be, log_n, batch, reps, mad_peak, check=True):
    """Fr NTT (natural order in and out) and the coset quotient by Z_Omega with data resident: times,
    roofline views against the measured multiply-add peak and the HBM peak, and the quotient's result
    checked on inputs whose quotient is known."""
    n = 1 << log_n
    out = {}
    data = synthetic_scalars(n * batch, 7)
    d = be.to_device(data)
    be.fr_ntt_dev(d, log_n, batch, False)
    be.fr_ntt_dev(d, log_n, batch, True)
    be.sync()
    fwd_ms = gpu_time(be, lambda: be.fr_ntt_dev(d, log_n, batch, False), reps)
    inv_ms = gpu_time(be, lambda: be.fr_ntt_dev(d, log_n, batch, True), reps)
    round_trip = None
    if check:
        # the timed loops ran forward and inverse transforms in different numbers (the warm-up is bounded by
        # time): the check starts again from the input, one forward and one inverse transform
        be.free(d)
        d = be.to_device(data)
        be.fr_ntt_dev(d, log_n, batch, False)
        be.fr_ntt_dev(d, log_n, batch, True)
        round_trip = bool(np.array_equal(be.from_device(d, (n * batch, 4)), data))
    single_ms = gpu_time(be, lambda: be.fr_ntt_dev(d, log_n, 1, False), 4 * reps)
    be.free(d)

    elems = n * batch
    mads = 0.5 * log_n * FR_MADS_PER_MUL  # per element
    traffic, src = measured_traffic("k_ntt_pass", "ntt_%dx2^%d" % (batch, log_n))

    def view(ms, mads_per_elem, bytes_per_elem, tr, tr_src):
        tm = elems * mads_per_elem / (ms * 1e-3)
        gb = elems * bytes_per_elem / (ms * 1e-3) / 1e9
        return {"ms": round(ms, 3), "bound": "valu_int32_mad",
                "valu": {"achieved": round(tm / 1e12, 3), "peak": round(mad_peak / 1e12, 3), "unit": "Tmad/s",
                         "frac": round(tm / mad_peak, 4), "mads_per_element": mads_per_elem},
                "hbm": {"achieved": round(gb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gb / HBM_PEAK_GBS, 4), "bytes_per_element": bytes_per_elem},
                "traffic": tr, "traffic_note": tr_src}

    out["ntt_roofline"] = view(fwd_ms, mads, 128, traffic, src)
    out["ntt_roofline"]["workload"] = "%d forward transforms of 2^%d points, natural order in and out" % (batch, log_n)
    out["ntt_roofline"]["inverse_ms"] = round(inv_ms, 3)
    out["ntt_roofline"]["single_transform_ms"] = round(single_ms, 4)
    out["ntt_roofline"]["round_trip_restores_input"] = round_trip

    # quotient: f_b = q_b * (X^n - 1), i.e. f_j = -q_j, f_(j+n) = q_j; the exact quotient is q_b
    q = synthetic_scalars(n * batch, 8)
    d_q = be.to_device(q)
    d_f = be.malloc(2 * n * batch * 32)
    d_work = be.malloc(2 * n * batch * 32)
    d_flags = be.malloc(4 * batch + 64)
    for b in range(batch):
        src_b = d_q + 32 * n * b
        be.fr_linear_combination_dev(src_b, n, None, 0, FR_MINUS_ONE, FR_ZERO, FR_ZERO, False, d_f + 32 * 2 * n * b)
        be.memcpy_d2d(d_f + 32 * (2 * n * b + n), src_b, 32 * n)

    def restore():  # the quotient works in place
        be.memcpy_d2d(d_work, d_f, 2 * n * batch * 32)

    restore()
    be.quotient_coset_dev(d_work, log_n, batch, d_flags)
    be.sync()
    q_ms = gpu_time(be, lambda: be.quotient_coset_dev(d_work, log_n, batch, d_flags), reps, before=restore)
    ok = None
    if check:
        got = be.from_device(d_work, (batch, 2 * n, 4))[:, :n, :].reshape(-1, 4)
        flags = be.from_device(d_flags, (batch,), dtype=np.int32)
        ok = bool(np.array_equal(got, q) and flags.all())
        # one polynomial that X^n - 1 does not divide: f_0 + 1 (an integer step on the residue: still below r
        # or, with negligible probability, not; either way a different field element)
        bad = be.from_device(d_f, (2 * n, 4))
        bad[0, 0] = np.uint64((int(bad[0, 0]) + 1) & 0xFFFFFFFFFFFFFFFF) if int(bad[0, 0]) != 0xFFFFFFFFFFFFFFFF \
            else np.uint64(int(bad[0, 0]) - 1)
        be.to_device(bad, d_work)
        be.quotient_coset_dev(d_work, log_n, 1, d_flags)
        be.sync()
        ok = ok and int(be.from_device(d_flags, (1,), dtype=np.int32)[0]) == 0
    for p in (d_q, d_f, d_work, d_flags):
        be.free(p)
    # per output coefficient: 2 transforms of 2 elements each on the doubled domain + 3 element-wise
    # passes of 2 elements; 512 bytes with the element-wise passes fused into the transforms
    q_mads = 2 * 2 * 0.5 * (log_n + 1) * FR_MADS_PER_MUL + 3 * 2 * FR_MADS_PER_MUL
    q_traffic, q_src = measured_traffic("k_ntt_pass", "quotient_%dx2^%d" % (batch, log_n))
    out["quotient_roofline"] = view(q_ms, q_mads, 512, q_traffic, q_src)
    out["quotient_roofline"]["workload"] = ("%d coset quotients by X^(2^%d) - 1, 2^%d -> 2^%d coefficients each, "
                                            "f = q * Z with q uniform" % (batch, log_n, log_n + 1, log_n))
    out["quotient_roofline"]["quotient_recovered_exactly_and_non_divisible_input_flagged"] = ok
    return out


def time_small_msm(log_n, reps, mad_peak):
    """The second MSM size of the headline metric on its own context: scalars resident, Mpts/s."""
    import plonk_amd

    be = plonk_amd.Backend(device=0)
    n = 1 << log_n
    be.kzg_setup(np.array([3, 1, 4, 1], dtype=np.uint64), n - 1, want_points=False)
    d_sc = be.to_device(synthetic_scalars(n, 2020))
    d_out = be.malloc(144)
    be.kzg_commit_dev(d_sc, n, d_out)
    be.sync()
    ms = gpu_time(be, lambda: be.kzg_commit_dev(d_sc, n, d_out), reps)
    # the same commitment eight times in one call: the serial tails run side by side
    for _ in range(4):  # clock warm-up, as in gpu_time
        be.kzg_commit_many_dev([(d_sc, n)] * 8)
    t0 = time.perf_counter()
    for _ in range(3):
        be.kzg_commit_many_dev([(d_sc, n)] * 8)
    batch_ms = (time.perf_counter() - t0) / 24 * 1e3
    be.profile(True)
    be.kzg_commit_dev(d_sc, n, d_out)
    phases = be.msm_phase_ms()
    plan = be.msm_plan()
    be.profile(False)
    adversarial = time_adversarial(be, log_n, np.array([3, 1, 4, 1], dtype=np.uint64), reps, uniform_ms=ms, d_buf=d_sc)
    be.free(d_sc)
    be.free(d_out)
    be.close()
    work = MADS_PER_POINT * n
    return {
        "adversarial_2^%d" % log_n: adversarial,
        "g1_msm_2^%d_ms" % log_n: round(ms, 3),
        "g1_msm_2^%d_mpts_per_s" % log_n: round(n / ms / 1e3, 1),
        "g1_msm_2^%d_in_a_batch_of_8_ms" % log_n: round(batch_ms, 3),
        "g1_msm_2^%d_in_a_batch_of_8_mpts_per_s" % log_n: round(n / batch_ms / 1e3, 1),
        "g1_msm_2^%d_roofline" % log_n: {
            "bound": "valu_int32_mad", "unit": "Tmad/s", "peak": round(mad_peak / 1e12, 3),
            "whole_msm": {"achieved": round(work / (ms * 1e-3) / 1e12, 3), "frac": round(work / (ms * 1e-3) / mad_peak, 4)},
            "in_a_batch_of_8": {"achieved": round(work / (batch_ms * 1e-3) / 1e12, 3),
                                "frac": round(work / (batch_ms * 1e-3) / mad_peak, 4)},
            "k_accumulate": {"ms": round(phases["accumulate"], 3),
                             "frac": round(work / (phases["accumulate"] * 1e-3) / mad_peak, 4)},
            "windows": plan["windows"], "phase_ms": {k: round(v, 3) for k, v in phases.items()},
            "note": "canonical 48000 multiply-adds per point / time of the whole commitment (all kernels)",
        },
    }


VERIFY_FOLD_SECONDS = None
VERIFY_RUN_SECONDS = None


def time_prover(log_gates, device_index=0, ranks=None, barrier=None):
    """Prove latency: the whole prover (plonk_amd/prover.py: 6 commitments, 15 openings, transforms,
    products and quotients) for a synthetic circuit of 2^log_gates - 1 gates, polynomials resident in
    device memory.  ranks: plonk_amd.sharding.TaskExchange when several GPUs share the proof's independent
    commitments and openings.  Returns (seconds per proof, seconds per proof with host buffers in and out
    of every call (single GPU only, else None), checks passed)."""
    import plonk_amd
    from plonk_amd import prover as pv

    be = plonk_amd.Backend(device=device_index)
    circuit = pv.synthetic_circuit(log_gates, 7)
    p = pv.Prover(be, circuit.d)
    tau = 0x123456789ABCDEF0123
    p.setup(tau)
    keys = p.keys(circuit)
    p.kzg_log = []
    proof = p.prove_resident(circuit, keys, ranks=ranks)  # warm-up, recorded for the checks below
    ok = p.audit_with_tau(tau) >= 2 and p.check_identities((circuit.number_gates, circuit.pub_inputs), proof, keys)
    # the verifier's pairing checks, folded into one (host code; only this rank's openings are in the log)
    tau_g2 = plonk_amd.kzg_setup_tau_g2(pv.to_mont([tau])[0])
    checks = p.opening_checks(known=[(keys["S"], keys["com_S"]), (keys["W"], keys["com_W"])])
    ok = ok and plonk_amd.kzg_verify_batch(tau_g2, *checks)  # (the first call also prepares the G2 side)
    t0 = time.perf_counter()
    ok = ok and plonk_amd.kzg_verify_batch(tau_g2, *checks)
    global VERIFY_FOLD_SECONDS, VERIFY_RUN_SECONDS
    VERIFY_FOLD_SECONDS = time.perf_counter() - t0
    if ranks is None:
        # the whole verifier on the serialised proof (verifier::run): decode, challenges, field equations, fold
        from plonk_amd import verifier
        vk = {"com_S": p.hex_of(keys["com_S"]), "com_W": p.hex_of(keys["com_W"])}
        vf = verifier.Verifier(be, circuit.d, circuit.number_gates, len(circuit.pub_inputs), tau_g2)
        ok = ok and vf.run(vk, proof)
        t0 = time.perf_counter()
        ok = ok and vf.run(vk, proof)
        VERIFY_RUN_SECONDS = time.perf_counter() - t0
    p.kzg_log = None
    if barrier:
        barrier()
    t0 = time.perf_counter()
    same = p.prove_resident(circuit, keys, ranks=ranks) == proof
    if barrier:
        barrier()
    seconds = time.perf_counter() - t0
    host_seconds = None
    if ranks is None:
        t0 = time.perf_counter()
        same = same and p.prove(circuit, keys) == proof
        host_seconds = time.perf_counter() - t0
    be.close()
    return seconds, host_seconds, bool(ok and same)


def run_transform_workload(args, be, world, rank, dist, torch, dev, backend):
    """--workload ntt / quotient: polynomial b of the batch on rank b mod N, no data-path collective."""
    from plonk_amd import sharding

    log_n, batch = args.ntt_log_n, args.ntt_batch
    n = 1 << log_n
    mine = sharding.polynomials_of_rank(batch, world, rank)
    quotient = args.workload == "quotient"
    per = 2 * n if quotient else n
    host = np.concatenate([synthetic_scalars(per, 500 + b) for b in mine]) if mine else np.zeros((0, 4), np.uint64)
    d_src = be.to_device(host) if mine else None
    d = be.malloc(max(len(mine), 1) * per * 32)
    d_flags = be.malloc(4 * batch + 64)

    def step():
        if not mine:
            return
        if quotient:
            be.memcpy_d2d(d, d_src, len(mine) * per * 32)  # the quotient overwrites its input
            be.quotient_coset_dev(d, log_n, len(mine), d_flags)
        else:
            be.fr_ntt_dev(d, log_n, len(mine), False)

    def fence():
        be.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        be.sync()
        torch.cuda.synchronize()

    if mine and not quotient:
        be.memcpy_d2d(d, d_src, len(mine) * per * 32)
    # the shader clock comes up over about 60 ms of continuous load (see gpu_time): steps of one or two
    # milliseconds need that before the warm-up steps proper, or they are timed on a rising clock
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < CLOCK_WARM_MS:
        step()
        be.sync()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        peak = be.bench_mad_rate()
        if quotient:
            mads = 2 * 2 * 0.5 * (log_n + 1) * FR_MADS_PER_MUL + 3 * 2 * FR_MADS_PER_MUL
            bytes_per = 512
        else:
            mads, bytes_per = 0.5 * log_n * FR_MADS_PER_MUL, 128
        local = len(mine) * n  # rank 0's own share against rank 0's GPU
        rec = {
            "metric": "coset_quotient_batch_ms" if quotient else "fr_ntt_batch_ms",
            "value": round(ms, 4), "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": ("%d coset quotients by X^(2^%d) - 1 (2^%d coefficients in, 2^%d out)"
                                    % (batch, log_n, log_n + 1, log_n)) if quotient else
                                   "%d forward Fr transforms of 2^%d points, natural order" % (batch, log_n),
                       "parallelism": "by polynomial x%d, no collective" % world,
                       "polynomials_on_rank_0": len(mine)},
            "roofline": {"bound": "valu_int32_mad", "kernel": "k_ntt_pass",
                         "achieved": round(local * mads / (ms * 1e-3) / 1e12, 3), "peak": round(peak / 1e12, 3),
                         "unit": "Tmad/s", "frac": round(local * mads / (ms * 1e-3) / peak, 4),
                         "traffic": None,
                         "hbm_view": {"achieved": round(local * bytes_per / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                      "unit": "GB/s", "frac": round(local * bytes_per / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                         "note": "%.0f multiply-adds and %d bytes per %s (rank 0's share / step time; the quotient's "
                                 "step includes restoring its input, 64 bytes per input coefficient)" % (
                                     mads, bytes_per, "output coefficient" if quotient else "element")},
        }
        print(json.dumps(rec), flush=True)
    be.free(d)
    be.free(d_flags)
    if d_src:
        be.free(d_src)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["msm", "ntt", "quotient"], default="msm")
    ap.add_argument("--log-n", type=int, default=24, help="log2 of the MSM size (whole job)")
    ap.add_argument("--ntt-log-n", type=int, default=20)
    ap.add_argument("--ntt-batch", type=int, default=16)
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--exchange", choices=["points", "buckets"], default="points",
                    help="what the ranks exchange: reduced partial points (144 B each) or bucket partials")
    ap.add_argument("--cpu-sample-log", type=int, default=19)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--prover-log-gates", type=int, default=20,
                    help="size of the synthetic circuit of the prove-latency extra (0: skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (see the docstring)" % args.gpus)
        args.gpus = world

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the product path has no CPU fallback")
    # PLONK_BENCH_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than
    # ranks (ranks then share devices and the 144-byte partials are staged through the host)
    backend = os.environ.get("PLONK_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import plonk_amd

    be = plonk_amd.Backend(device=device_index, window_bits=args.window_bits)
    if args.workload != "msm":
        run_transform_workload(args, be, world, rank, dist, torch, dev, backend)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        be.close()
        return

    n_total = 1 << args.log_n
    assert n_total % world == 0
    n_local = n_total // world
    first = rank * n_local

    # --- set-up (not timed): SRS slice + window tables resident, coefficients resident
    # (tau's limbs are used as a Montgomery residue throughout: set-up and check alike)
    tau = np.array([0x0123456789ABCDEF, 0x0FEDCBA987654321, 0x1122334455667788, 0x0102030405060708],
                   dtype=np.uint64)
    t_setup = time.perf_counter()
    be.kzg_setup_range(tau, first, n_local)
    be.sync()
    t_setup = time.perf_counter() - t_setup
    scalars = synthetic_scalars(n_local, 1000 + rank)
    d_scalars = be.to_device(scalars)
    local_pt = torch.zeros(18, dtype=torch.int64, device=dev)
    gathered = torch.zeros(18 * world, dtype=torch.int64, device=dev)
    total_pt = torch.zeros(18, dtype=torch.int64, device=dev)
    be.profile(True)
    phases = []

    from plonk_amd import sharding

    nbuckets, each = be.msm_bucket_layout()
    send = recv = None
    if world > 1:
        send = torch.empty(nbuckets * each, dtype=torch.uint8, device=dev)
        recv = torch.empty(nbuckets * each, dtype=torch.uint8, device=dev)

    def step(record, buckets):
        if buckets:
            sharding.commit_sharded_buckets(be, d_scalars, n_local, send, recv, local_pt, gathered, total_pt)
        else:
            sharding.commit_sharded(be, d_scalars, n_local, local_pt, gathered, total_pt)
        if record:
            phases.append(be.msm_phase_ms())  # waits for this step's MSM events

    def fence():
        be.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        be.sync()
        torch.cuda.synchronize()

    def timed(buckets, record):
        for _ in range(args.warmup):
            step(False, buckets)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(record, buckets)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    use_buckets = args.exchange == "buckets" and world > 1
    elapsed = timed(use_buckets, True)
    # the other exchange as well, so that a multi-GPU run records both; a failure there is a note in the
    # record, not the loss of the headline figure
    other_elapsed = None
    if world > 1:
        try:
            other_elapsed = timed(not use_buckets, False)
        except Exception as exc:  # noqa: BLE001 -- reported in the bench line
            other_elapsed = "failed: %r" % (exc,)

    # the timed commitment against the known answer [f(tau)]G, formed by the library itself: f(tau) by the
    # evaluation kernel, then a one-coefficient commitment (SRS point 0 is G)
    self_check = None
    if world == 1:
        f_tau = be.fr_eval(scalars, tau)
        want = be.kzg_commit(np.ascontiguousarray(f_tau).reshape(1, 4))
        be.sync()
        got = local_pt.cpu().numpy().view(np.uint64)  # one rank: the local result is the commitment
        enc = be.g1_compress(np.stack([np.asarray(want).reshape(18), got.reshape(18)]))
        self_check = bool(enc[0] == enc[1])

    # prove latency with the proof's independent commitments and openings dealt out to the ranks: every
    # rank takes part; the tiny result exchange runs over its own gloo group with a short timeout, so that
    # a failure on one rank ends in an error note instead of a stuck job
    shared_prover = None
    if world > 1 and args.prover_log_gates and not args.no_extras:
        import faulthandler
        try:
            import datetime
            faulthandler.dump_traceback_later(75, exit=False)  # a stuck rank shows where it is
            group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=90))
            ranks = sharding.TaskExchange(group=group)
            sec, _, checked = time_prover(args.prover_log_gates, device_index, ranks,
                                          lambda: dist.barrier(group=group))
            t = torch.tensor([sec, 1.0 if checked else 0.0], dtype=torch.float64)
            dist.all_reduce(t[0:1], op=dist.ReduceOp.MAX, group=group)
            dist.all_reduce(t[1:2], op=dist.ReduceOp.MIN, group=group)
            shared_prover = (float(t[0]), bool(t[1] > 0.5))
            faulthandler.cancel_dump_traceback_later()
        except Exception as exc:  # noqa: BLE001 -- reported in the bench line
            import traceback
            traceback.print_exc()
            faulthandler.cancel_dump_traceback_later()
            shared_prover = "failed: %r" % (exc,)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_total * args.steps / elapsed / 1e6
        plan = be.msm_plan()
        acc_ms = float(np.mean([p["accumulate"] for p in phases]))
        peak = be.bench_mad_rate()
        achieved = MADS_PER_POINT * n_local / (acc_ms * 1e-3)
        issued_per_point = plan["windows"] * FQ_MULS_PER_ADDITION * MADS_PER_FQ_MUL
        issued = issued_per_point * n_local / (acc_ms * 1e-3)
        phase_avg = {k: round(float(np.mean([p[k] for p in phases])), 3) for k in phases[0]}
        traffic, traffic_src = measured_traffic("k_accumulate", "msm_2^%d_x%d" % (args.log_n, world))
        rec = {
            "metric": "g1_msm_mpts_per_s",
            "value": round(value, 2),
            "unit": "Mpts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": "one 2^%d-point BLS12-381 G1 MSM (KZG commit), synthetic powers-of-tau SRS, "
                            "sharded by point range" % args.log_n,
                "log_n": args.log_n,
                "points_per_gpu": n_local,
                "window_bits": plan["window_bits"],
                "windows": plan["windows"],
                "buckets": plan["buckets"],
                "parallelism": ("point-range x%d, all_to_all of bucket partials (%d B per rank)"
                                % (world, nbuckets * each)) if use_buckets else
                               "point-range x%d, all_gather of %d-byte partials" % (world, 144),
                "srs_setup_s": round(t_setup, 2),
            },
            "roofline": {
                "bound": "valu_int32_mad",
                "kernel": "k_accumulate",
                "achieved": round(achieved / 1e12, 3),
                "peak": round(peak / 1e12, 3),
                "unit": "Tmad/s",
                "frac": round(achieved / peak, 4),
                "traffic": traffic,
                "traffic_note": traffic_src,
                "issued": {"mads_per_point": issued_per_point, "achieved": round(issued / 1e12, 3),
                           "frac": round(issued / peak, 4)},
                "note": "achieved = canonical algorithmic work, %d multiply-adds per point (%d windows x %d Fq "
                        "multiplications x %d), x points per launch / average k_accumulate duration (HIP events on "
                        "the library stream); the plan run here uses %d windows of %d bits over pre-shifted copies "
                        "of the SRS, i.e. issues %d multiply-adds per point: `issued` prices the same kernel time "
                        "at that figure; peak = v_mad_u64_u32 register-only microbenchmark run in this process, timed "
                        "after 80 ms of the same kernel, i.e. at the shader clock the part sustains (records taken "
                        "before that correction show peaks of 28-31 Tmad/s measured on a rising clock and fractions "
                        "about a tenth higher for the same kernel times)"
                        % (MADS_PER_POINT, CANONICAL_WINDOWS, FQ_MULS_PER_ADDITION, MADS_PER_FQ_MUL, plan["windows"],
                           plan["window_bits"], issued_per_point),
                "kernel_ms": round(acc_ms, 3),
                "whole_step_frac": round(MADS_PER_POINT * n_local / (ms_per_step * 1e-3) / peak, 4),
                "hbm_view": hbm_view(be, n_local, plan["windows"], acc_ms),
            },
            "msm_phase_ms": phase_avg,
        }
        if self_check is not None:
            rec["result_equals_f_of_tau_times_G"] = self_check
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            rec["cpu_baseline"] = cpu_baseline(be, n_local, args.cpu_sample_log, threads, args.log_n)
        extras = {}
        if other_elapsed is not None:
            extras["exchange_%s_ms_per_step" % ("buckets" if use_buckets else "points")] = round(ms_per_step, 3)
            extras["exchange_%s_ms_per_step" % ("points" if use_buckets else "buckets")] = (
                other_elapsed if isinstance(other_elapsed, str) else round(other_elapsed / args.steps * 1e3, 3))
        if world == 1 and not args.no_extras:
            be.profile(False)
            # through host buffers, as `kzg_commit(&gp, &f)` hands them over (PCIe-inclusive; never `value`)
            be.kzg_commit(scalars)
            t0 = time.perf_counter()
            for _ in range(3):
                be.kzg_commit(scalars)
            extras["g1_msm_2^%d_host_buffers_ms" % args.log_n] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
            # the copy alone (pageable host memory to the resident buffer, synchronous), for the split of that figure
            be.to_device(scalars, d_scalars)
            t0 = time.perf_counter()
            for _ in range(3):
                be.to_device(scalars, d_scalars)
            be.sync()
            extras["h2d_2^%d_scalars_ms" % args.log_n] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
            # the benchmark's own commitment four times in one call (tails run side by side)
            be.kzg_commit_many_dev([(d_scalars, n_local)] * 2)
            t0 = time.perf_counter()
            be.kzg_commit_many_dev([(d_scalars, n_local)] * 4)
            extras["g1_msm_2^%d_in_a_batch_of_4_ms" % args.log_n] = round((time.perf_counter() - t0) / 4 * 1e3, 3)
            extras.update(time_small_msm(20, 10, peak))
            # scalar sets that skew the bucket method, at both sizes of the headline metric, each against [f(tau)]G
            extras["adversarial"] = {
                "2^20": extras.pop("adversarial_2^20"),
                "2^%d" % args.log_n: time_adversarial(be, args.log_n, tau, 3, uniform_ms=ms_per_step, d_buf=d_scalars),
                "note": "one resident commitment per set (every scalar equal, every scalar r - 1, one in sixteen zero, "
                        "only 0 and 1, all zero), ms by HIP events; vs_uniform: against uniform scalars of the same size "
                        "on the same context"}
            be.to_device(scalars, d_scalars)
            extras.update(transform_rooflines(be, 20, 16, 5, peak))
            extras["fr_ntt_2^20_ms"] = round(extras["ntt_roofline"]["ms"] / 16, 4)
            if not args.no_cpu_baseline:
                extras["cpu_baseline_transforms"] = cpu_baseline_ntt(min(16, os.cpu_count() or 1))
            if args.prover_log_gates:
                prove_s, prove_host_s, checked = time_prover(args.prover_log_gates)
                extras["prove_latency_s"] = round(prove_s, 3)
                extras["prove_latency_host_buffers_s"] = round(prove_host_s, 3)
                extras["prove_latency_note"] = (
                    "full proof (6 commitments, 15 openings) of a synthetic circuit of 2^%d - 1 gates, d = 3*2^%d rows, "
                    "1 GPU; polynomials resident in device memory (trace, S, W uploaded once per proof) / every call "
                    "through host buffers; both give the same proof, whose commitments and openings match the test "
                    "SRS trapdoor, which satisfies the verifier's field equations and whose openings pass the "
                    "verifier's pairing checks (kzg_verify, folded into one): %s" % (
                        args.prover_log_gates, args.prover_log_gates, checked))
                if VERIFY_FOLD_SECONDS is not None:
                    extras["verify_openings_folded_ms"] = round(VERIFY_FOLD_SECONDS * 1e3, 1)
                    if VERIFY_RUN_SECONDS is not None:
                        extras["verify_latency_ms"] = round(VERIFY_RUN_SECONDS * 1e3, 1)
                    extras["verify_openings_note"] = (
                        "the proof's openings checked by pairings as ONE pairing-product check (host code: two Miller "
                        "loops, one final exponentiation, two small G1 multi-scalar sums), commitments already at hand; "
                        "verify_latency_ms: the whole verifier on the serialised proof (plonk_amd/verifier.py: decoding, "
                        "the commitment of the public-input polynomial, challenges, field equations, the fold)")
        if shared_prover is not None:
            if isinstance(shared_prover, tuple):
                extras["prove_latency_s"] = round(shared_prover[0], 3)
                extras["prove_latency_note"] = (
                    "full proof of a synthetic circuit of 2^%d - 1 gates; the 18 independent commitments and openings are "
                    "dealt out by size to the %d GPUs (sharing by polynomial), every rank builds only the polynomials its "
                    "own tasks need, and the results are gathered; each rank's share checked against the test SRS trapdoor, verifier's field equations: %s"
                    % (args.prover_log_gates, world, shared_prover[1]))
            else:
                extras["prove_latency_s"] = None
                extras["prove_latency_note"] = shared_prover
        if extras:
            rec["extras"] = extras
        print(json.dumps(rec), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    be.close()


if __name__ == "__main__":
    main()
