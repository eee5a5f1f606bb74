"""Development probe: throughput of the per-net drop-in calls (the compatibility path)."""
import time
import numpy as np
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
orc = rc.load_oracle()
text = sc.synthetic_text(5000)
for S in (1, 8, 64):
    a = sc.ApiSet(amd, softmax_best_guess=orc.orc_softmax_best_guess, input_size=42, hidden_size=1024,
                  output_size=42, S=S, D=20, learn_rate=1e-5, seed=1)
    for i in range(25):
        a.char_step(text, i, rc.WEIGHTED, 0.95)
    amd.rnn_amd_synchronize()
    n = 20
    t0 = time.time()
    for i in range(25, 25 + n):
        a.char_step(text, i, rc.WEIGHTED, 0.95)
    amd.rnn_amd_synchronize()
    dt = time.time() - t0
    print("per-net path: S=%d  %.0f stream-timesteps/s  (%.0f us per stream-step)" % (S, n * S / dt, 1e6 * dt / (n * S)))
    a.close()
