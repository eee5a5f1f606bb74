"""Development probe: the callers' device paths (multi-head text step, gstclassify order with class groups and
an active mask, rnnca's dense inputs with the sigmoid-slope loss) on RANDOM shapes against the oracle, from a
cold start for a few generations each: gpu_stress_callers.py <seed> <trials> [multi|classify|rnnca].
A trial whose hidden masks differ at the end (a pre-activation within rounding of zero) is reported as a flip,
not as a mismatch."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import recur_ctypes as rc
import replay
import scenarios as sc

amd = rc.load_amd()
rc.bind_char(amd)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 12
which = sys.argv[3] if len(sys.argv) > 3 else "all"
rs = np.random.default_rng(seed)
KEYS = ["ih_w", "ho_w", "ih_m", "ho_m", "ih_delta", "ho_delta", "hidden", "output", "hist", "o_error",
        "min_error_factor", "ih_scale"]
bad = 0
LR = float(os.environ.get("LR_SCALE", "1"))  # hotter regimes: clipped error gains, early exits


def verdict(label, g, o, exact):
    global bad
    sg, so = g.snapshot(), o.snapshot()
    flips = int(((sg["hidden"] != 0) != (so["hidden"] != 0)).sum())
    flips += int(((sg["hist"] != 0) != (so["hist"] != 0)).sum())  # ... or in a slot of the history ring
    try:
        replay.check(sg, so, 2e-4, keys=KEYS, exact=exact, elementwise=False)
        res = "ok"
    except AssertionError as e:
        res = "MISMATCH " + str(e)[:int(os.environ.get("STRESS_MSG", "160"))]
        if flips == 0:
            bad += 1
    print("%s flips %d: %s" % (label, flips, res), flush=True)
    g.close()
    o.close()


def multi(trial):
    A = int(rs.integers(2, 81))
    NC = int(rs.integers(1, max(2, min(60, 3900 // A))))
    hidden = int(rs.choice([24, 40, 99, 128, 256, 512]))
    S = int(rs.integers(1, 41)) if hidden >= 256 else int(rs.integers(1, 80))
    D = int(rs.integers(2, 8))
    leak = float(rs.choice([0.0, 0.3, 0.8]))
    noise = float(rs.choice([0.0, 0.02]))
    act = int(rs.choice([rc.RELU, rc.RESQRT]))
    method = int(rs.choice([rc.WEIGHTED, rc.NESTEROV, rc.WEIGHTED]))  # ADAGRAD from a cold start without ballast is 0 / 0 on both sides
    kw = dict(input_size=A, hidden_size=hidden, output_size=A * NC, S=S, D=D, learn_rate=LR * (1e-3 if hidden < 256 else 1e-4),
              seed=300 + trial, noise=noise, activation=act)
    print("next: multi", kw, "leak", leak, "method", method, flush=True)
    g, o = sc.AmdBatchedSet(amd, **kw), sc.OracleSet(**kw)
    ranges = (C.c_int * (2 * (NC + 1)))()
    for step in range(D + 3):
        hot = rs.integers(0, A, S).astype(np.int32)
        nxt = rs.integers(0, A, S).astype(np.int32)
        cls = rs.integers(0, NC, S).astype(np.int32)
        amd.rnn_amd_set_multi_step_deltas(g.handle, rc.iptr(hot), rc.iptr(nxt), rc.iptr(cls), A, leak, 0)
        amd.rnn_apply_learning(g.net, method, 0.9)
        for j in range(S):
            o.orc.orc_advance(o.z, j)
            o.orc.orc_multi_softmax_error(o.z, j, int(hot[j]), int(nxt[j]), int(cls[j]), A, leak, ranges)
            o.orc.orc_calc_deltas(o.z, j, 1 if j else 0, ranges)
        o.orc.orc_apply_learning(o.z, method, 0.9)
    verdict("multi A%d NC%d h%d S%d D%d leak%.1f noise%.2f act%d m%d" % (A, NC, hidden, S, D, leak, noise, act, method),
            g, o, ("index", "generation", "rng"))


def classify(trial):
    ng = int(rs.integers(1, 5))
    gsize = rs.integers(2, 7, ng).astype(np.int32)
    goff = np.concatenate([[0], np.cumsum(gsize)[:-1]]).astype(np.int32)
    O = int(gsize.sum())
    NIN = int(rs.integers(1, 70))
    hidden = int(rs.choice([32, 48, 128, 256, 512]))
    S = int(rs.integers(1, 150)) if hidden <= 256 else int(rs.integers(1, 70))
    D = int(rs.integers(2, 9))
    method = int(rs.choice([rc.WEIGHTED, rc.NESTEROV]))
    kw = dict(input_size=NIN, hidden_size=hidden, output_size=O, S=S, D=D, learn_rate=LR * (1e-3 if hidden < 256 else 1e-4),
              seed=500 + trial)
    print("next: classify", kw, list(gsize), flush=True)
    g, o = sc.AmdBatchedSet(amd, **kw), sc.OracleSet(**kw)
    weight = (0.5 + rs.random(O)).astype(np.float32)
    wins, wrong = C.c_int(0), C.c_float(0)
    for step in range(D + 3):
        x = (rs.standard_normal((S, NIN)) * 0.6).astype(np.float32)
        targets = np.stack([rs.integers(-1, gs, S) for gs in gsize], axis=1).astype(np.int32)
        if rs.random() < 0.3:
            targets[rs.integers(0, S)] = -1
        targets = np.ascontiguousarray(targets)
        trained = np.zeros(S, np.uint8)
        use_w = bool(step % 2)
        amd.rnn_bptt_clear_deltas(g.net)
        amd.rnn_amd_set_opinion(g.handle, rc.fptr(x), NIN, None)
        amd.rnn_amd_set_grouped_softmax_error(g.handle, ng, rc.iptr(goff), rc.iptr(gsize), rc.iptr(targets),
                                              rc.fptr(weight) if use_w else None, rc.u8ptr(trained))
        amd.rnn_amd_set_calc_deltas(g.handle, 1, None, rc.u8ptr(trained))
        amd.rnn_amd_set_advance(g.handle)
        amd.rnn_apply_learning(g.net, method, 0.9)
        o.orc.orc_clear_deltas(o.z)
        for j in range(S):
            o.orc.orc_opinion(o.z, j, rc.fptr(np.ascontiguousarray(x[j])), 0.0)
            n = o.orc.orc_grouped_softmax_error(o.z, j, ng, rc.iptr(goff), rc.iptr(gsize),
                                                rc.iptr(np.ascontiguousarray(targets[j])),
                                                rc.fptr(weight) if use_w else None, C.byref(wins), C.byref(wrong))
            assert bool(n) == bool(trained[j]), ("trained mask", step, j)
            if n:
                o.orc.orc_calc_deltas(o.z, j, 1, None)
            o.orc.orc_advance(o.z, j)
        o.orc.orc_apply_learning(o.z, method, 0.9)
    verdict("classify groups%s in%d h%d S%d D%d m%d" % (list(gsize), NIN, hidden, S, D, method), g, o,
            ("index", "generation"))


def rnnca(trial):
    NIN = int(rs.choice([35, 35, 12, 3]))
    NO = int(rs.integers(1, 6))
    hidden = int(rs.choice([32, 64, 128, 256, 512]))
    S = int(rs.integers(1, 300)) if hidden <= 128 else int(rs.integers(1, 100))
    D = int(rs.integers(1, 7))
    kw = dict(input_size=NIN, hidden_size=hidden, output_size=NO, S=S, D=D, learn_rate=LR * (1e-3 if hidden < 256 else 1e-4),
              seed=700 + trial, momentum=0.95)
    print("next: rnnca", kw, flush=True)
    g, o = sc.AmdBatchedSet(amd, **kw), sc.OracleSet(**kw)
    for gen in range(D + 3):
        x = np.ascontiguousarray(rs.random((S, NIN)).astype(np.float32))
        tgt = np.ascontiguousarray(rs.random((S, NO)).astype(np.float32))
        m = amd.rnn_calculate_momentum_soft_start(float(gen), 0.95, 2000.0)
        amd.rnn_bptt_clear_deltas(g.net)
        amd.rnn_amd_set_advance(g.handle)
        amd.rnn_amd_set_opinion(g.handle, rc.fptr(x), NIN, None)
        amd.rnn_amd_set_sigmoid_mse_error(g.handle, rc.fptr(tgt), NO, NO)
        amd.rnn_amd_set_calc_deltas(g.handle, 1, None, None)
        amd.rnn_apply_learning(g.net, rc.WEIGHTED, m)
        o.orc.orc_clear_deltas(o.z)
        for j in range(S):
            o.orc.orc_advance(o.z, j)
            o.orc.orc_opinion(o.z, j, rc.fptr(np.ascontiguousarray(x[j])), 0.0)
            o.orc.orc_sigmoid_mse_error(o.z, j, rc.fptr(np.ascontiguousarray(tgt[j])), NO)
            o.orc.orc_calc_deltas(o.z, j, 1, None)
        o.orc.orc_apply_learning(o.z, rc.WEIGHTED, m)
    verdict("rnnca in%d out%d h%d S%d D%d" % (NIN, NO, hidden, S, D), g, o, ("index", "generation"))


for trial in range(trials):
    for name, fn in (("multi", multi), ("classify", classify), ("rnnca", rnnca)):
        if which in ("all", name):
            fn(trial)
print("bad (without mask flips):", bad)
