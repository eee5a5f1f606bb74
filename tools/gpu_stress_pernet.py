"""Development probe: the per-net calls on random small shapes (k_fwd_small, k_bptt_small, the padded
one-launch chain) against the oracle: gpu_stress_pernet.py <seed> <trials>."""
import sys, os
sys.path.insert(0, "tests")
import numpy as np, recur_ctypes as rc, scenarios as sc, replay
amd = rc.load_amd(); orc = rc.load_oracle()
rs = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
text = sc.synthetic_text(6000)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    hidden = int(rs.choice([int(rs.integers(8, 125)), int(rs.integers(8, 125)), 256, 512]))
    S = int(rs.integers(1, 5)); D = int(rs.integers(2, 12)); act = int(rs.choice([rc.RELU, rc.RESQRT, rc.RECLIP20]))
    isz = int(rs.choice([42, 7, 60]))
    kw = dict(input_size=isz, hidden_size=hidden, output_size=isz, S=S, D=D, learn_rate=1e-3 if hidden < 200 else 1e-4, seed=200 + trial, activation=act)
    g = sc.ApiSet(amd, softmax_best_guess=orc.orc_softmax_best_guess, **kw); o = sc.OracleSet(**kw)
    t2 = np.asarray(text) % isz
    for i in range(D + 4):
        g.char_step(t2, i, rc.WEIGHTED, 0.9); o.char_step(t2, i, rc.WEIGHTED, 0.9)
    sg, so = g.snapshot(), o.snapshot()
    flips = int(((sg["hidden"] != 0) != (so["hidden"] != 0)).sum()) + int(((sg["hist"] != 0) != (so["hist"] != 0)).sum())
    try:
        replay.check(sg, so, 2e-4, keys=["ih_w", "ho_w", "ih_delta", "ho_delta", "hidden", "hist", "min_error_factor", "ih_scale", "output", "o_error"], exact=("index", "generation"), elementwise=False)
        res = "ok"
    except AssertionError as e:
        res = "MISMATCH " + str(e)[:150]
        if flips == 0: bad += 1
    print("h%d i%d S%d D%d act%d flips %d: %s" % (hidden, isz, S, D, act, flips, res))
    g.close(); o.close()
print("bad (without mask flips):", bad)
