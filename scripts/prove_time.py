"""Wall time of the whole prover (plonk_amd/prover.py) for a synthetic circuit of 2^k - 1 gates on one
GPU; host buffers in and out of every call, so host copies are included.
usage: python scripts/prove_time.py LOG_GATES [REPS]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd
from plonk_amd import prover as pv

k = int(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
be = plonk_amd.Backend(0)
t = time.perf_counter()
circuit = pv.synthetic_circuit(k, 7)
t_circ = time.perf_counter() - t
p = pv.Prover(be, circuit.d)
t = time.perf_counter()
p.setup(0x123456789ABCDEF0123)
be.sync()
t_setup = time.perf_counter() - t
t = time.perf_counter()
keys = p.keys(circuit)
t_keys = time.perf_counter() - t
print("gates 2^%d - 1, d = %d: circuit %.2f s, SRS + window tables (degree %d) %.2f s, keys %.3f s, plan %s"
      % (k, circuit.d, t_circ, 2 * circuit.d - 3, t_setup, t_keys, be.msm_plan()), flush=True)
for rep in range(reps):
    times = {}
    t = time.perf_counter()
    proof = p.prove(circuit, keys, times)
    total = time.perf_counter() - t
    print("prove %.3f s  " % total + "  ".join("%s %.3f" % kv for kv in times.items()), flush=True)
host_proof = proof
for rep in range(reps):
    times = {}
    t = time.perf_counter()
    proof = p.prove_resident(circuit, keys, times)
    total = time.perf_counter() - t
    print("prove_resident %.3f s  " % total + "  ".join("%s %.3f" % kv for kv in times.items()), flush=True)
print("same proof as the host-buffer path:", proof == host_proof, flush=True)
print("com_T", proof["com_T"][:32], "output", proof["output"][:24], flush=True)
# one more proof with every commitment and opening recorded, checked against the trapdoor of this
# test SRS; then the verifier's field equations on the opened values
p.kzg_log = []
proof = p.prove_resident(circuit, keys)
t = time.perf_counter()
items = p.audit_with_tau(0x123456789ABCDEF0123)
ok = p.check_identities((circuit.number_gates, circuit.pub_inputs), proof, keys)
print("audit: %d commitments and openings match the trapdoor evaluation; verifier equations hold: %s (%.2f s)"
      % (items, ok, time.perf_counter() - t), flush=True)
