"""Development probe: rnn_char_epoch's multi-tap branch (charmodel-predict.c:260-420) on random nets, set sizes, texts,
start / stop positions, optimisers and momentum soft starts, against the oracle's per-stream loop with the same
momentum schedule: gpu_stress_epoch.py <seed> <trials>"""
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
    A = int(rs.integers(2, 60))
    hidden = int(rs.choice([9, 40, 99, 128, 256]))
    S = int(rs.integers(1, 40))
    single = bool(rs.random() < 0.35)  # the single-net branch: rnn_bptt_calculate per symbol, temporal batching
    if single:
        S = 1
    batch = int(rs.choice([1, 1, 3, 5])) if single else 1
    D = int(rs.integers(1, 7))
    L = int(rs.integers(S + 3, S + 120))
    start = int(rs.integers(0, max(1, L - 3)))
    n_gen = int(rs.integers(1, 14))
    use_stop = bool(rs.integers(0, 2))
    method = rc.WEIGHTED if single else int(rs.choice([rc.WEIGHTED, rc.NESTEROV, rc.SIMPLIFIED_NESTEROV, rc.CLASSICAL]))
    soft = float(rs.choice([0.0, 50.0, 2000.0]))
    noise = float(rs.choice([0.0, 0.02]))
    kw = dict(input_size=A, hidden_size=hidden, output_size=A, S=S, D=D, learn_rate=1e-3, seed=int(rs.integers(1, 1000)),
              activation=int(rs.choice([rc.RELU, rc.RESQRT])), noise=noise)
    print("next:", kw, "len", L, "start", start, "single net, batch %d," % batch if single else "multi-tap,", "generations", n_gen, "stop" if use_stop else "to the end", "method", method,
          "soft start", soft, flush=True)
    text = np.ascontiguousarray(sc.synthetic_text(8000, alphabet=A)[:L])
    a = sc.ApiSet(amd, **kw)
    o = sc.OracleSet(**kw)
    model = rc.CharModel()
    model.net = a.net
    model.training_nets = a.nets
    model.n_training_nets = S
    model.batch_size = batch
    model.momentum = 0.9
    model.momentum_soft_start = soft
    model.learning_style = method
    model.report_interval = int(rs.choice([3, 100]))
    model.save_net = False
    model.use_multi_tap_path = not single
    amd.rnn_char_init_schedule(C.byref(model.schedule), 0, 0.0, 1.0, 0)
    v = rc.CharVentropy()
    amd.rnn_char_init_ventropy(C.byref(v), a.net, rc.u8ptr(text), 0, 1)  # no validation text
    stop = n_gen if use_stop else 0
    done = amd.rnn_char_epoch(C.byref(model), None, C.byref(v), rc.u8ptr(text), L, start, stop, 0.0, 0, -1, 2, 0, 0)
    gen = 0
    want_done = 0
    for i in range(start, L - 1):
        m = amd.rnn_calculate_momentum_soft_start(float(gen), 0.9, soft)
        if single:
            o.orc.orc_advance(o.z, 0)
            o.orc.orc_net_error_bptt(o.z, 0, int(text[i]), int(text[i + 1]), C.byref(C.c_int(0)))
            o.orc.orc_bptt_calculate(o.z, 0, batch, m)
        else:
            o.char_step(text, i, method, m)
        gen += 1
        if stop and gen >= stop:
            want_done = 1
            break
    res = "ok"
    sg, so = a.snapshot(), o.snapshot()
    flips = int(((sg["hidden"] != 0) != (so["hidden"] != 0)).sum()) + int(((sg["hist"] != 0) != (so["hist"] != 0)).sum())
    try:
        assert done == want_done, ("return value", done, want_done)
        assert a.net.contents.generation == gen, ("generations", a.net.contents.generation, gen)
        replay.check(sg, so, 2e-4, keys=["ih_w", "ho_w", "ih_m", "ho_m", "ih_delta", "ho_delta", "hidden", "hist",
                                         "min_error_factor", "ih_scale"], exact=("index", "generation", "rng"), elementwise=False)
    except AssertionError as e:
        res = "MISMATCH " + str(e)[:300]
        if flips == 0:
            bad += 1
    print("   flips %d: %s" % (flips, res), flush=True)
    amd.rnn_char_delete_ventropy(C.byref(v))
    a.close()
    o.close()
print("bad (without mask flips):", bad)
