"""Development probe: steady-state rates of the callers' own workloads at BASELINE.json's sizes
(configs[2] gstclassify training generation, configs[3] multi-head step, configs[4] rnnca training generation).  Under tools/kstats-style
profiling: rocprofv3 --kernel-trace --stats -- python3 gpu_callers_rate.py classify|multi|rnnca"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
which = sys.argv[1] if len(sys.argv) > 1 else "both"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
if which in ("multi", "both"):
    A, NC, D, S = 73, 50, 20, int(os.environ.get("TUNE_S", "256"))
    g = sc.AmdBatchedSet(amd, input_size=A, hidden_size=1024, output_size=A * NC, S=S, D=D, learn_rate=1e-4, seed=61,
                         activation=rc.RESQRT, noise=float(os.environ.get("TUNE_NOISE", "0.01")), flags=rc.FLAG_STANDARD | rc.FLAG_ADAPTIVE_MIN_ERROR)
    amd.rnn_set_momentum_values(g.net, 200.0)
    rs = np.random.default_rng(3)
    draws = [(rs.integers(0, A, S).astype(np.int32), rs.integers(0, A, S).astype(np.int32),
              rs.integers(0, NC, S).astype(np.int32)) for _ in range(8)]
    def gen(i):
        hot, nxt, cls = draws[i % 8]
        if os.environ.get("RATE_TWO_CALLS"):  # the deltas, then rnn_apply_learning's launch (until round 5 the only form)
            amd.rnn_amd_set_multi_step_deltas(g.handle, rc.iptr(hot), rc.iptr(nxt), rc.iptr(cls), A, 0.1, 0)
            amd.rnn_apply_learning(g.net, rc.ADAGRAD, 0.9)
        else:  # one call: the ADAGRAD update in the weight-delta GEMM's epilogue
            amd.rnn_amd_set_multi_step(g.handle, rc.iptr(hot), rc.iptr(nxt), rc.iptr(cls), A, 0.1, rc.ADAGRAD, 0.9)
    for i in range(D + 5):
        gen(i)
    amd.rnn_amd_synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        gen(i)
    amd.rnn_amd_synchronize()
    dt = time.perf_counter() - t0
    print("configs[3] multi-head generation 1024 / %d streams / 73 x 50 outputs: %.0f stream-timesteps/s, %.1f us per generation"
          % (S, n * S / dt, 1e6 * dt / n))
    g.close()
if which in ("rnnca", "both"):
    H, S, D = 2048, 512, 10
    g = sc.AmdBatchedSet(amd, input_size=35, hidden_size=H, output_size=3, S=S, D=D, learn_rate=1e-5, seed=81, momentum=0.95)
    rs = np.random.default_rng(11)
    xs = [np.ascontiguousarray((rs.integers(0, 256, (S, 35)) / np.float32(255.0)).astype(np.float32)) for _ in range(4)]
    ts = [np.ascontiguousarray((rs.integers(0, 256, (S, 3)) / np.float32(255)).astype(np.float32)) for _ in range(4)]
    def gen(i):
        m = amd.rnn_calculate_momentum_soft_start(float(i), 0.95, 2000.0)
        amd.rnn_bptt_clear_deltas(g.net)
        amd.rnn_amd_set_advance(g.handle)
        if os.environ.get("RATE_SEPARATE"):
            amd.rnn_amd_set_opinion(g.handle, rc.fptr(xs[i % 4]), 35, None)
            amd.rnn_amd_set_sigmoid_mse_error(g.handle, rc.fptr(ts[i % 4]), 3, 3)
        elif os.environ.get("RATE_TWO_CALLS"):
            amd.rnn_amd_set_opinion_sigmoid_mse(g.handle, rc.fptr(xs[i % 4]), 35, rc.fptr(ts[i % 4]), 3, 3)
        else:  # the generation as one call: the update rides in the delta GEMM's epilogue
            amd.rnn_amd_set_dense_step_sigmoid_mse(g.handle, rc.fptr(xs[i % 4]), 35, rc.fptr(ts[i % 4]), 3, 3, rc.WEIGHTED, m)
            return
        amd.rnn_amd_set_calc_deltas(g.handle, 1, None, None)
        amd.rnn_apply_learning(g.net, rc.WEIGHTED, m)
    for i in range(D + 5):
        gen(i)
    amd.rnn_amd_synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        gen(i)
    amd.rnn_amd_synchronize()
    dt = time.perf_counter() - t0
    print("configs[4] rnnca training generation 2048 / %d cells / depth %d: %.0f cell-timesteps/s, %.1f us per generation"
          % (S, D, n * S / dt, 1e6 * dt / n))
    g.close()
if which in ("classify", "both"):
    # gstclassify's train_channel order (gstclassify.c:2070-2130): opinion, class-group loss on the
    # device, calc_deltas over the channels that trained, advance; Nesterov update per generation
    H, S, D, NIN = 512, 128, 30, 32
    g = sc.AmdBatchedSet(amd, input_size=NIN, hidden_size=H, output_size=2, S=S, D=D, learn_rate=3e-4, seed=6)
    rs = np.random.default_rng(12)
    goff, gsize = np.array([0], np.int32), np.array([2], np.int32)
    xs = [np.ascontiguousarray((rs.standard_normal((S, NIN)) * 0.5).astype(np.float32)) for _ in range(4)]
    tg = [np.ascontiguousarray(rs.integers(-1, 2, (S, 1)).astype(np.int32)) for _ in range(4)]
    trained = np.zeros(S, np.uint8)
    def gen(i):
        amd.rnn_bptt_clear_deltas(g.net)
        if os.environ.get("RATE_SEPARATE"):
            amd.rnn_amd_set_opinion(g.handle, rc.fptr(xs[i % 4]), NIN, None)
            amd.rnn_amd_set_grouped_softmax_error(g.handle, 1, rc.iptr(goff), rc.iptr(gsize), rc.iptr(tg[i % 4]), None,
                                                  rc.u8ptr(trained))
        else:
            amd.rnn_amd_set_opinion_grouped_softmax(g.handle, rc.fptr(xs[i % 4]), NIN, 1, rc.iptr(goff), rc.iptr(gsize),
                                                    rc.iptr(tg[i % 4]), None, rc.u8ptr(trained))
        amd.rnn_amd_set_calc_deltas(g.handle, 1, None, rc.u8ptr(trained))
        amd.rnn_amd_set_advance(g.handle)
        amd.rnn_apply_learning(g.net, rc.NESTEROV, 0.9)
    for i in range(D + 5):
        gen(i)
    amd.rnn_amd_synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        gen(i)
    amd.rnn_amd_synchronize()
    dt = time.perf_counter() - t0
    print("configs[2] gstclassify training generation 512 / %d channels / depth %d: %.0f channel-timesteps/s, %.1f us per generation"
          % (S, D, n * S / dt, 1e6 * dt / n))
    g.close()
