"""Random commitments against the known answer [f(tau)]G (development aid; the answer needs no oracle: f(tau) by
the library's evaluation kernel, then a one-coefficient commitment, compared as compressed encodings).
usage: python scripts/fuzz_msm.py SECONDS [SEED [MULTIPLE [MAX_LOG]]]
Sizes 1 .. 2^18 (log-uniform), window widths 0 (automatic) or 2 .. 20, scalars uniform or from a small set of
special values (0, 1, r - 1, (r - 1) / 2, (r + 1) / 2, 2^254, repeated), single commitments, batches, and openings
checked by the pairing equation.  With MULTIPLE (e.g. 2048) the lengths are whole multiples of it where the SRS is
long enough: together with PLONK_HIP_FRONT_END=scalars every such commitment takes the sort's first pass from the scalars."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonk_amd
from plonk_amd.prover import R_MOD, to_mont

seconds = float(sys.argv[1])
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
multiple = int(sys.argv[3]) if len(sys.argv) > 3 else 1
max_log = int(sys.argv[4]) if len(sys.argv) > 4 else 18  # SRS sizes up to 2^MAX_LOG (log-uniform)
special = [0, 1, 2, R_MOD - 1, R_MOD - 2, (R_MOD - 1) // 2, (R_MOD + 1) // 2, 1 << 254, (1 << 254) - 1, (1 << 255) % R_MOD]
t0 = last = time.perf_counter()
cases = failures = 0
while time.perf_counter() - t0 < seconds:
    n = max(1, int(2 ** rnd.uniform(0, max_log)))
    c = rnd.choice([0, 0, 0] + list(range(2, 21))) if n <= 1 << 18 else rnd.choice([0, 0] + list(range(15, 24)))
    be = plonk_amd.Backend(0, window_bits=c)
    tau = to_mont([rnd.randrange(1, R_MOD)])[0]
    be.kzg_setup(tau, n - 1, want_points=False)
    tau_g2 = plonk_amd.kzg_setup_tau_g2(tau)
    for _ in range(4):
        m = rnd.randint(1, n)
        if m >= multiple:
            m -= m % multiple
        kind = rnd.random()
        if kind < 0.5:
            f = np.frombuffer(rnd.randbytes(m * 32), dtype=np.uint64).reshape(m, 4).copy()
            f[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
        elif kind < 0.8:
            f = to_mont([rnd.choice(special) for _ in range(m)])
        else:
            f = to_mont([rnd.choice(special[:4])] * m)
        got = [be.kzg_commit(f)]
        if rnd.random() < 0.3:
            got.append(be.kzg_commit_batch([f, f])[1])
        want = be.kzg_commit(np.ascontiguousarray(be.fr_eval(f, tau)).reshape(1, 4))
        enc = be.g1_compress(np.stack([np.asarray(want).reshape(18)] + [np.asarray(g).reshape(18) for g in got]))
        ok = all(e == enc[0] for e in enc[1:])
        if m >= 2 and rnd.random() < 0.5:
            u = to_mont([rnd.choice(special + [rnd.randrange(R_MOD)])])[0]
            v, proof = be.kzg_evaluate(f, u)
            ok = ok and np.array_equal(np.asarray(v).reshape(4), be.fr_eval(f, u))
            ok = ok and plonk_amd.kzg_verify(tau_g2, got[0], u, v, proof)  # the opening, by pairings
        cases += 1
        if not ok:
            failures += 1
            print("MISMATCH n=%d c=%d m=%d kind=%.2f" % (n, c, m, kind), flush=True)
    be.close()
    if time.perf_counter() - last > 20:
        last = time.perf_counter()
        print("%d cases so far, %d failures" % (cases, failures), flush=True)
print("%d cases, %d failures, %.0f s" % (cases, failures, time.perf_counter() - t0))
sys.exit(1 if failures else 0)
