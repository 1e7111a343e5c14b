import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import plonk_amd
from plonk_amd import prover as pv
be = plonk_amd.Backend(0)
k = 20
circuit = pv.synthetic_circuit(k, 7)
p = pv.Prover(be, circuit.d)
p.setup(0x1234567)
keys = p.keys(circuit)
d, ng, w = p.d, circuit.number_gates, p.w
def T(name, fn, reps=3):
    fn(); be.sync()
    t = time.perf_counter()
    for _ in range(reps): fn()
    be.sync()
    print("%-40s %7.2f ms" % (name, (time.perf_counter() - t) / reps * 1e3), flush=True)
n1 = d
a = be.to_device(np.ascontiguousarray(p.omega)); b = be.to_device(np.ascontiguousarray(p.omega[::-1]))
out = be.malloc((3 * d) * 32)
roots = be.to_device(np.ascontiguousarray(p.omega[0:3*ng:3]))
big = be.malloc((2*d + ng) * 32); be.memset_zero(big, (2*d+ng)*32); be.memcpy_d2d(big, a, d*32)
q = be.malloc((2*d) * 32)
one = pv._m(1)
T("upload d coefficients", lambda: be.to_device(p.omega, a))
T("ntt_mixed d", lambda: be.fr_ntt_mixed_dev(a, d, 1, True))
T("poly_mul d x d", lambda: be.fr_poly_mul_dev(a, d, b, d, out))
T("poly_mul ng x d", lambda: be.fr_poly_mul_dev(a, ng, b, d, out))
T("poly_mul ng x 2d", lambda: be.fr_poly_mul_dev(a, ng, out, 2*d-1, big))
T("div_by_roots (2d+ng) by ng roots", lambda: be.fr_div_by_roots_dev(big, 2*d+ng-2, roots, ng, q))
T("div_by_roots d by 2 roots", lambda: be.fr_div_by_roots_dev(a, d, roots, 2, q))
T("div_vanishing 2d by d", lambda: be.fr_div_vanishing_dev(big, 2*d-1, d, q))
T("scale_by_powers d", lambda: be.fr_scale_by_powers_dev(a, d, pv._m(w)))
T("linear_combination d", lambda: be.fr_linear_combination_dev(a, d, b, d, one, one, one, False, out))
T("ratio_prefix_product d", lambda: (be.memcpy_d2d(out, a, d*32), be.fr_ratio_prefix_product_dev(out, b, d)))
T("commit d", lambda: be.kzg_commit_many_dev([(a, d)]))
T("commit 2d-2", lambda: be.kzg_commit_many_dev([(big, 2*d-2)]))
T("commit ng", lambda: be.kzg_commit_many_dev([(a, ng)]))
T("open d", lambda: be.kzg_open_many_dev(a, d, pv.to_mont([12345])))
