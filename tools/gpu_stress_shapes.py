"""Development probe: random chain-eligible shapes (hidden 256 / 512 / 1024, any stream count, all activations)
against the oracle: gpu_stress_shapes.py <seed> <trials>."""
import sys, os
sys.path.insert(0, "tests")
import numpy as np, recur_ctypes as rc, scenarios as sc, replay
amd = rc.load_amd()
rs = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
text = sc.synthetic_text(6000)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    hidden = int(rs.choice([256, 512, 256, 1024]))
    S = int(rs.integers(1, 70)); D = int(rs.integers(2, 9)); act = int(rs.choice([rc.RELU, rc.RESQRT, rc.RECLIP20]))
    if hidden == 1024: S = int(rs.integers(1, 40)); D = int(rs.integers(2, 6))
    kw = dict(input_size=42, hidden_size=hidden, output_size=42, S=S, D=D, learn_rate=1e-4 if hidden < 1024 else 1e-5, seed=100 + trial, activation=act)
    g = sc.AmdBatchedSet(amd, **kw); o = sc.OracleSet(**kw)
    steps = D + 3
    ok = True
    for i in range(steps):
        g.char_step(text, i, rc.WEIGHTED, 0.9); o.char_step(text, i, rc.WEIGHTED, 0.9)
    sg, so = g.snapshot(), o.snapshot()
    flips = int(((sg["hidden"] != 0) != (so["hidden"] != 0)).sum()) + int(((sg["hist"] != 0) != (so["hist"] != 0)).sum())
    try:
        replay.check(sg, so, 2e-4, keys=["ih_w", "ho_w", "ih_delta", "ho_delta", "hidden", "hist", "min_error_factor", "ih_scale"], exact=("index", "generation"), elementwise=False)
        res = "ok"
    except AssertionError as e:
        res = "MISMATCH " + str(e)[:150]
        if flips == 0: bad += 1
    print("h%d S%d D%d act%d flips %d: %s" % (hidden, S, D, act, flips, res))
    g.close(); o.close()
print("bad (without mask flips):", bad)
