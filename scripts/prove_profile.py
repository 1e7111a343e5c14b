"""Host-side profile of one proof (development aid).  usage: python scripts/prove_profile.py LOG_GATES"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plonk_amd
from plonk_amd import prover as pv

k = int(sys.argv[1])
be = plonk_amd.Backend(0)
circuit = pv.synthetic_circuit(k, 7)
p = pv.Prover(be, circuit.d)
p.setup(0x123456789ABCDEF0123)
keys = p.keys(circuit)
p.prove(circuit, keys)
pr = cProfile.Profile()
pr.enable()
p.prove(circuit, keys)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
