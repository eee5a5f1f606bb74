"""Development probe (not a test): prints GPU-vs-oracle parity numbers for a
few shapes through both the per-net drop-in calls and the batched calls."""
import sys, time
import ctypes as C
import numpy as np
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import recur_ctypes as rc, scenarios as sc

amd = rc.load_amd(); orc = rc.load_oracle()
print(amd.rnn_amd_version().decode(), "devices", amd.rnn_amd_device_count(), flush=True)
text = sc.synthetic_text(20000)

def run(kind, hidden, S, D, steps, act=rc.RELU, method=rc.WEIGHTED, lr=1e-3):
    kw = dict(input_size=42, hidden_size=hidden, output_size=42, S=S, D=D, activation=act, learn_rate=lr, seed=3)
    o = sc.OracleSet(**kw)
    if kind == "pernet":
        a = sc.ApiSet(amd, softmax_best_guess=orc.orc_softmax_best_guess, **kw)
    else:
        a = sc.AmdBatchedSet(amd, **kw)
    if method == rc.ADAGRAD:
        amd.rnn_set_momentum_values(a.net, 0.1); o.arrays()['ih_m'][:] = 0.1; o.arrays()['ho_m'][:] = 0.1
    t0 = time.time()
    for i in range(steps):
        a.char_step(text, i, method, 0.9)
    amd.rnn_amd_synchronize(); t1 = time.time()
    for i in range(steps):
        o.char_step(text, i, method, 0.9)
    t2 = time.time()
    sa, so = a.snapshot(), o.snapshot()
    errs = {k: rc.rel_err(sa[k], so[k]) for k in sc.FLOAT_KEYS}
    ex = {k: bool(np.array_equal(sa[k], so[k])) for k in sc.EXACT_KEYS}
    print(f"{kind:8s} H={hidden} S={S} D={D} act={act} m={method} steps={steps} gpu {t1-t0:.3f}s cpu {t2-t1:.3f}s", flush=True)
    print("   ", " ".join(f"{k}={v:.2e}" for k, v in errs.items()), ex, "depth", so['bptt_depth'][:4], flush=True)
    a.close(); o.close()

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "small"):
    run("pernet", 39, 4, 8, 12)
    run("batched", 39, 4, 8, 12)
    run("batched", 39, 4, 8, 12, act=rc.RESQRT, method=rc.NESTEROV)
    run("batched", 39, 4, 8, 12, act=rc.RECLIP20, method=rc.ADAGRAD)
    run("pernet", 99, 3, 10, 6, act=rc.RESQRT)
    run("batched", 99, 70, 10, 6)
if which in ("all", "big"):
    run("batched", 1024, 64, 20, 3, lr=1e-5)
