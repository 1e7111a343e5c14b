"""Random transforms, products and quotients against independent kernels of the same library (development aid).
usage: python scripts/fuzz_ntt.py SECONDS [SEED]
  * fr_ntt / fr_ntt_mixed (n = 2^k or 3 * 2^k, k up to 18, batches, coset shifts): value j of the transform must be
    f(shift * w^j), checked at random j by the Horner evaluation kernel; the inverse must give f back exactly.
  * fr_poly_mul: (a b)(x) = a(x) b(x) at a random x.
  * quotient_coset: f = q (X^k - 1) built by hand gives q back and the divisible flag; f + 1 is flagged.
Everything is exact integer arithmetic: any difference is an error.  FUZZ_BREAK=1 in the environment flips a bit of
every forward transform before it is checked: the run must then report mismatches (a check of the checker)."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonk_amd
from plonk_amd.prover import R_MOD, from_mont, root_of_unity, to_mont

seconds = float(sys.argv[1])
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
be = plonk_amd.Backend(0)


def rand_poly(n):
    f = np.frombuffer(rnd.randbytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    f[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
    return f


def ev(f, x):
    return from_mont(be.fr_eval(f, to_mont([x])[0]))[0]


t0 = last = time.perf_counter()
cases = failures = 0
while time.perf_counter() - t0 < seconds:
    kind = rnd.random()
    ok = True
    what = ""
    if kind < 0.45:
        k = rnd.randint(1, 18)
        batch = rnd.choice([1, 1, 2, 3, 5, 16]) if k <= 14 else 1
        n = 1 << k
        shift = rnd.choice([None, None, 7, rnd.randrange(1, R_MOD)])
        f = rand_poly(n * batch).reshape(batch, n, 4)
        sh = None if shift is None else to_mont([shift])[0]
        vals = be.fr_ntt(f, False, sh)
        if os.environ.get("FUZZ_BREAK"):
            vals[0, 0, 0] ^= np.uint64(1)
        w = root_of_unity(n)
        for _ in range(3):
            b, j = rnd.randrange(batch), rnd.randrange(n)
            x = (shift or 1) * pow(w, j, R_MOD) % R_MOD
            ok = ok and from_mont(vals[b, j:j + 1])[0] == ev(f[b], x)
        ok = ok and np.array_equal(be.fr_ntt(vals, True, sh), f)
        what = "ntt k=%d batch=%d shift=%s" % (k, batch, shift)
    elif kind < 0.6:
        k = rnd.randint(0, 16)
        n = 3 << k
        f = rand_poly(n)
        vals = be.fr_ntt_mixed(f, False)
        w = root_of_unity(n)
        for _ in range(3):
            j = rnd.randrange(n)
            ok = ok and from_mont(vals[j:j + 1])[0] == ev(f, pow(w, j, R_MOD))
        ok = ok and np.array_equal(be.fr_ntt_mixed(vals, True), f)
        what = "mixed n=3*2^%d" % k
    elif kind < 0.8:
        na, nb = int(2 ** rnd.uniform(0, 17)), int(2 ** rnd.uniform(0, 17))
        a, b = rand_poly(na), rand_poly(nb)
        c = be.fr_poly_mul(a, b)
        x = rnd.randrange(R_MOD)
        ok = c.shape[0] == na + nb - 1 and ev(c, x) == ev(a, x) * ev(b, x) % R_MOD
        what = "mul %d x %d" % (na, nb)
    else:
        k = rnd.randint(3, 16)
        n = 1 << k
        batch = rnd.choice([1, 2, 3])
        q = rand_poly(n * batch).reshape(batch, n, 4)
        f = np.zeros((batch, 2 * n, 4), dtype=np.uint64)
        minus_q = to_mont([(R_MOD - v) % R_MOD for v in from_mont(q.reshape(-1, 4))]).reshape(batch, n, 4)
        f[:, :n] = minus_q
        f[:, n:] = q
        got, flags = be.quotient_coset(f)
        ok = np.array_equal(got, q) and all(bool(x) for x in flags)
        g = f.copy()
        g[0, 0] = to_mont([(from_mont(g[0, 0:1])[0] + 1) % R_MOD])[0]
        _, flags = be.quotient_coset(g)
        ok = ok and not bool(flags[0]) and all(bool(x) for x in flags[1:])
        what = "quotient k=%d batch=%d" % (k, batch)
    cases += 1
    if not ok:
        failures += 1
        print("MISMATCH", what, flush=True)
    if time.perf_counter() - last > 20:
        last = time.perf_counter()
        print("%d cases so far, %d failures (last: %s)" % (cases, failures, what), flush=True)
print("%d cases, %d failures, %.0f s" % (cases, failures, time.perf_counter() - t0))
sys.exit(1 if failures else 0)
