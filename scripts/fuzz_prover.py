"""Random small circuits through both paths of the prover against the CPU restatement of the reference's prover
(development aid): keys and proof must be equal as JSON.  Circuits the reference itself cannot prove (its own
assertions or panics, raised by the restatement as well) are skipped and counted.
usage: python scripts/fuzz_prover.py SECONDS [SEED]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonk_amd
from oracle import pyref as o
from plonk_amd import prover as pv

seconds = float(sys.argv[1])
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
be = plonk_amd.Backend(0)
t0 = time.perf_counter()
cases = skipped = failures = 0
while time.perf_counter() - t0 < seconds:
    log_gates = rnd.randint(2, 5)
    full = (1 << log_gates) - 1
    number_gates = rnd.choice([None, None, rnd.randint(2, full)])
    seed = rnd.randrange(1 << 30)
    tau = rnd.randrange(1, o.R)
    try:
        oc = o.synthetic_circuit(log_gates, seed, number_gates=number_gates)
        d = len(oc.trace)
        setup = o.Setup(len(oc.gates) if number_gates is None else number_gates, 2, 1, d,
                        o.kzg_setup_with_tau(2 * d - 3, tau))
        pk, vk = o.reproduce_proving_key(setup, oc), o.reproduce_verification_key(setup, oc)
        want = o.reproduce_proof(setup, pk, vk, oc)
    except (AssertionError, ZeroDivisionError, IndexError, ValueError, OverflowError, RuntimeError) as exc:
        skipped += 1
        continue
    cases += 1
    circuit = pv.synthetic_circuit(log_gates, seed, number_gates=number_gates)
    p = pv.Prover(be, d)
    p.setup(tau)
    try:
        keys = p.keys(circuit)
        ok = {"com_S": p.hex_of(keys["com_S"]), "com_W": p.hex_of(keys["com_W"])} == vk
        ok = ok and p.prove(circuit, keys) == want and p.prove_resident(circuit, keys) == want
    except Exception as exc:  # noqa: BLE001 -- reported with the case
        ok = False
        print("exception", repr(exc), flush=True)
    if not ok:
        failures += 1
        print("FAIL log_gates %d number_gates %s seed %d tau %d" % (log_gates, number_gates, seed, tau), flush=True)
    if cases % 20 == 0:
        print("%d cases so far, %d skipped, %d failures" % (cases, skipped, failures), flush=True)
print("%d cases, %d skipped, %d failures, %.0f s" % (cases, skipped, failures, time.perf_counter() - t0), flush=True)
sys.exit(1 if failures else 0)
