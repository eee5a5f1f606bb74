"""Development probe: the batched text step on ARBITRARY shapes (any hidden size, alphabet, stream count, depth,
activation, optimiser, with and without noise) against the oracle from a cold start:
gpu_stress_text.py <seed> <trials>.  A trial whose zero masks differ at the end (hidden layer or a slot of the
history) is reported as a flip, not as a mismatch."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import recur_ctypes as rc
import replay
import scenarios as sc

amd = rc.load_amd()
rs = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    hidden = int(rs.choice([int(rs.integers(3, 700)), int(rs.choice([64, 128, 256, 384, 512, 640, 768, 1024]))]))
    A = int(rs.integers(2, 130))
    big = hidden > 300
    S = int(rs.integers(1, 60 if big else 200))
    D = int(rs.integers(1, 8 if big else 25))
    act = int(rs.choice([rc.RELU, rc.RESQRT, rc.RECLIP20]))
    method = int(rs.choice([rc.WEIGHTED, rc.NESTEROV, rc.CLASSICAL]))
    noise = float(rs.choice([0.0, 0.0, 0.03]))
    kw = dict(input_size=A, hidden_size=hidden, output_size=A, S=S, D=D,
              learn_rate=float(rs.choice([1e-3, 1e-4])) / (4 if big else 1), seed=900 + trial, activation=act, noise=noise)
    print("next:", kw, "method", method, flush=True)
    text = sc.synthetic_text(8000, alphabet=A)
    i0 = 0
    if os.environ.get("SHORT_TEXT"):  # texts barely longer than the set is wide: the streams' offsets wrap
        L = int(rs.integers(S + 2, S + 60))
        text = np.ascontiguousarray(text[:L])
        i0 = int(rs.integers(0, max(1, L - 1 - (D + 4))))
    g, o = sc.AmdBatchedSet(amd, **kw), sc.OracleSet(**kw)
    for i in range(i0, min(i0 + D + 4, len(text) - 1)):
        g.char_step(text, i, method, 0.9)
        o.char_step(text, i, method, 0.9)
    sg, so = g.snapshot(), o.snapshot()
    flips = int(((sg["hidden"] != 0) != (so["hidden"] != 0)).sum()) + int(((sg["hist"] != 0) != (so["hist"] != 0)).sum())
    try:
        replay.check(sg, so, 2e-4, keys=["ih_w", "ho_w", "ih_m", "ho_m", "ih_delta", "ho_delta", "hidden", "output", "hist",
                                         "o_error", "min_error_factor", "ih_scale"], exact=("index", "generation", "rng"), elementwise=False)
        res = "ok"
    except AssertionError as e:
        res = "MISMATCH " + str(e)[:200]
        if flips == 0:
            bad += 1
    print("h%d A%d S%d D%d act%d m%d noise%.2f flips %d: %s" % (hidden, A, S, D, act, method, noise, flips, res), flush=True)
    g.close()
    o.close()
print("bad (without mask flips):", bad)
