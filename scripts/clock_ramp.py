"""How long the shader clock takes to come up after the GPU has been idle: per-call times of a fixed kernel
sequence (16 transforms of 2^20 points) started after a pause (development aid)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonk_amd

be = plonk_amd.Backend(0)
n, batch = 1 << 20, 16
data = np.frombuffer(np.random.default_rng(1).bytes(n * batch * 32), dtype=np.uint64).reshape(-1, 4).copy()
data[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)
d = be.to_device(data)
be.fr_ntt_dev(d, 20, batch, False)
be.sync()
for pause in (2.0, 0.2, 0.02):
    time.sleep(pause)
    ms = []
    for _ in range(120):
        be.timer_start()
        be.fr_ntt_dev(d, 20, batch, False)
        ms.append(be.timer_stop())
    acc = np.cumsum(ms)
    print("after %.2f s idle: calls 1-5 %s; call 10 %.3f, 20 %.3f, 40 %.3f, 80 %.3f, 120 %.3f ms; elapsed at call 10/20/40: %.0f/%.0f/%.0f ms"
          % (pause, " ".join("%.3f" % v for v in ms[:5]), ms[9], ms[19], ms[39], ms[79], ms[119], acc[9], acc[19], acc[39]), flush=True)
