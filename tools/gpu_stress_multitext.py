"""Development probe: the multi-head trainer (rnn_char_multitext_train / _spin / rnn_char_multi_cross_entropy with the
reference's signatures, charmodel-multi-predict.c:74-408) on random nets, alphabets, head counts, texts, batch sizes,
optimisers, leakage and noise, against the oracle's restatement of the loop: gpu_stress_multitext.py <seed> <trials>"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import recur_ctypes as rc
import replay
import scenarios as sc

amd = rc.bind_char(rc.load_amd())
rs = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    A = int(rs.integers(3, 60))
    NC = int(rs.integers(1, 12))
    hidden = int(rs.choice([17, 40, 99, 128, 256]))
    D = int(rs.integers(2, 9))
    method = int(rs.choice([rc.ADAGRAD, rc.WEIGHTED, rc.NESTEROV]))
    batch = int(rs.choice([1, 2, 5, 7, 16]))
    leakage = float(rs.choice([0.0, 0.3, 0.8]))
    noise = float(rs.choice([0.0, 0.02]))
    act = int(rs.choice([rc.RELU, rc.RESQRT]))
    kw = dict(input_size=A, hidden_size=hidden, output_size=A * NC, S=1, D=D, learn_rate=float(rs.choice([1e-3, 3e-3])),
              seed=int(rs.integers(1, 1000)), activation=act, noise=noise, momentum=0.9)
    print("next:", kw, "NC", NC, "method", method, "batch", batch, "leak", leakage, flush=True)
    g = sc.ApiSet(amd, **kw)
    o = sc.OracleSet(**kw)
    if method == rc.ADAGRAD:
        amd.rnn_set_momentum_values(g.net, 0.5)
        a = o.arrays()
        a["ih_m"][:] = 0.5
        a["ho_m"][:] = 0.5
    report = rc.CharProgressReport()
    res = "ok"
    try:
        for _ in range(int(rs.integers(1, 4))):
            cls, n = int(rs.integers(0, NC)), int(rs.integers(2, 150))
            text = np.ascontiguousarray(rs.integers(0, A, n).astype(np.uint8))
            amd.rnn_char_multitext_train(g.net, rc.u8ptr(text), n, A, cls, leakage, C.byref(report), None, method, 0.123,
                                         batch, None, None, None, 0)
            e, h = C.c_float(0), C.c_float(0)
            o.orc.orc_multitext_train(o.z, 0, rc.u8ptr(text), n, A, cls, leakage, method, 0.9, batch, C.byref(e), C.byref(h))
            assert abs(report.training_error - e.value / (n - 1)) <= 2e-4 * abs(e.value / (n - 1)) + 1e-6, "training error"
            assert abs(report.training_entropy + h.value / (n - 1)) <= 2e-4 * abs(h.value / (n - 1)) + 1e-6, "training entropy"
        sg, so = g.snapshot(), o.snapshot()
        flips = int(((sg["hidden"] != 0) != (so["hidden"] != 0)).sum()) + int(((sg["hist"] != 0) != (so["hist"] != 0)).sum())
        # (up to three texts of up to 150 symbols with batched updates in between: free-running state, so the norm metrics --
        # the element-wise reading of the bar is for comparisons one generation deep, profiles/NOTES_r05.md section 4)
        replay.check(sg, so, 2e-4, keys=["ih_w", "ho_w", "ih_m", "ho_m", "ih_delta", "ho_delta", "hidden", "hist",
                                         "min_error_factor"], exact=("index", "generation", "rng"), elementwise=False)
        n = int(rs.integers(1, 12))
        text = np.ascontiguousarray(rs.integers(0, A, n).astype(np.uint8))
        amd.rnn_char_multitext_spin(g.net, rc.u8ptr(text), n, None, None, None, 0)
        for i in range(n):
            o.orc.orc_advance(o.z, 0)
            o.orc.orc_one_hot_opinion(o.z, 0, int(text[i]), noise)
        sg, so = g.snapshot(), o.snapshot()
        replay.check(sg, so, 2e-4, keys=["hidden", "output", "hist", "ih_w", "ho_w"], exact=("index", "generation", "rng"), elementwise=False)
        n = int(rs.integers(8, 80))
        text = np.ascontiguousarray(rs.integers(0, A, n).astype(np.uint8))
        ent_g = (C.c_double * NC)(*([0.0] * NC))
        ent_o = (C.c_double * NC)(*([0.0] * NC))
        ign = int(rs.integers(0, min(7, n - 2)))
        amd.rnn_char_multi_cross_entropy(g.net, rc.u8ptr(text), n, A, ent_g, ign)
        o.orc.orc_multi_cross_entropy(o.z, 0, rc.u8ptr(text), n, A, ent_o, ign)
        for j in range(NC):
            assert abs(ent_g[j] - ent_o[j]) <= 2e-4 * abs(ent_o[j]), ("cross entropy of head", j, ent_g[j], ent_o[j])
        sg, so = g.snapshot(), o.snapshot()
        flips += int(((sg["hidden"] != 0) != (so["hidden"] != 0)).sum())
        replay.check(sg, so, 2e-4, keys=["hidden", "output", "hist"], exact=("index", "generation", "rng"), elementwise=False)
    except AssertionError as e:
        sg, so = g.snapshot(), o.snapshot()
        flips = int(((sg["hidden"] != 0) != (so["hidden"] != 0)).sum()) + int(((sg["hist"] != 0) != (so["hist"] != 0)).sum())
        res = "MISMATCH " + str(e)[:300]
        if flips == 0:
            bad += 1
    print("   flips %d: %s" % (flips, res), flush=True)
    g.close()
    o.close()
print("bad (without mask flips):", bad)
