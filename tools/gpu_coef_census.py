"""Development probe (GPU box): how often the weight-delta GEMM's per-(step, stream) coefficient is exactly 1.0
at the north-star shape -- ih_scale == 1 (no soft clip of the error sum, recur-nn.c:393-404) for a stream
whose BPTT ran all its steps.  Decides whether a "this K tile's coefficients are all ones" fast path in
k_delta_dma is the common case."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
text = sc.synthetic_text(60000)
g = sc.AmdBatchedSet(amd, input_size=42, hidden_size=1024, output_size=42, S=256, D=20, learn_rate=1e-5, seed=1)
g.load_text(text)
i = 0
for block in range(12):
    for _ in range(50):
        amd.rnn_amd_set_char_step(g.handle, i, rc.WEIGHTED, 0.95)
        i += 1
    s = g.snapshot()["ih_scale"]
    print("after %4d generations: ih_scale == 1 for %3d of %d streams; min %.4f" % (i, int((s == 1.0).sum()), s.size, float(s.min())))
