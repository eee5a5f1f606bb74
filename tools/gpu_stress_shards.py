"""Development probe: the sharded (multi-GPU) path emulated on ONE GPU for random shapes and rank counts: R shards of
S streams each (rnn_amd_new_training_set_shard + rnn_amd_set_shard), each computing its deltas into its own external
device buffer, the buffers summed by hand (what the all-reduce leaves on every rank), rnn_apply_learning on each replica
-- against the oracle's single set of all R x S streams; replicas bit-identical, generator states exact.
    gpu_stress_shards.py <seed> <trials>"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import recur_ctypes as rc
import replay
import scenarios as sc

amd = rc.load_amd()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipFree.argtypes = [C.c_void_p]
D2H, H2D = 2, 1
rs = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    R = int(rs.choice([2, 2, 3, 4, 8]))
    hidden = int(rs.choice([24, 45, 64, 128, 256, 512, 1024]))
    S = int(rs.choice([1, 3, 16, 32, 33])) if hidden < 1024 else int(rs.choice([16, 32]))
    if hidden >= 512:
        R = min(R, 3)
    D = int(rs.integers(2, 7 if hidden <= 256 else 4))
    A = int(rs.integers(5, 60))
    noise = float(rs.choice([0.0, 0.03]))
    act = int(rs.choice([rc.RELU, rc.RESQRT]))
    method = int(rs.choice([rc.WEIGHTED, rc.NESTEROV]))
    kw = dict(input_size=A, hidden_size=hidden, output_size=A, D=D, learn_rate=1e-3 if hidden < 512 else 1e-4,
              seed=int(rs.integers(1, 1000)), activation=act)
    print("next: %d ranks x %d streams %s noise %.2f method %d" % (R, S, kw, noise, method), flush=True)
    text = sc.synthetic_text(6000, alphabet=A)
    ranks = [sc.AmdBatchedSet(amd, S=S, noise=noise, shard=(r * S, R * S), **kw) for r in range(R)]
    n_delta = ranks[0].I * ranks[0].H + ranks[0].H * ranks[0].O
    bufs = []
    for r, g in enumerate(ranks):
        g.load_text(text)
        amd.rnn_amd_set_shard(g.handle, r * S, R * S)
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), 4 * n_delta) == 0
        amd.rnn_amd_set_external_delta(g.handle, p)
        bufs.append(p)
    o = sc.OracleSet(S=R * S, noise=noise, **kw)
    h = [np.zeros(n_delta, np.float32) for _ in range(R)]
    for i in range(D + 3):
        for g in ranks:
            amd.rnn_amd_set_char_step_deltas(g.handle, i)
        amd.rnn_amd_synchronize()
        for r in range(R):
            assert hip.hipMemcpy(h[r].ctypes.data, bufs[r], 4 * n_delta, D2H) == 0
        total = h[0].copy()
        for r in range(1, R):
            total += h[r]
        for r, g in enumerate(ranks):
            assert hip.hipMemcpy(bufs[r], total.ctypes.data, 4 * n_delta, H2D) == 0
            amd.rnn_apply_learning(g.net, method, 0.9)
        o.char_step(text, i, method, 0.9)
    snaps = [g.snapshot() for g in ranks]
    so = o.snapshot()
    res = "ok"
    try:
        for k in ("ih_w", "ho_w", "ih_m", "ho_m"):
            for r in range(1, R):
                assert np.array_equal(snaps[0][k], snaps[r][k]), "replicas differ in " + k
        both = {k: np.concatenate([s_[k] for s_ in snaps], axis=1 if k == "hist" else 0)
                for k in ("hist", "hidden", "output", "o_error", "min_error_factor", "ih_scale", "index", "generation", "rng")}
        flips = int(((both["hidden"] != 0) != (so["hidden"] != 0)).sum()) + int(((both["hist"] != 0) != (so["hist"] != 0)).sum())
        replay.check(snaps[0], so, 2e-4, keys=["ih_w", "ho_w", "ih_m", "ho_m"], exact=(), elementwise=False)
        replay.check(snaps[R - 1], so, 2e-4, keys=["ih_delta", "ho_delta"], exact=(), elementwise=False)
        replay.check(both, so, 2e-4, keys=["hist", "hidden", "output", "o_error", "min_error_factor", "ih_scale"],
                     exact=("index", "generation", "rng"), elementwise=False)
    except AssertionError as e:
        res = "MISMATCH " + str(e)[:300]
        if flips == 0 or "replicas" in str(e):
            bad += 1
    print("   flips %d: %s" % (flips, res), flush=True)
    for r, g in enumerate(ranks):
        amd.rnn_amd_set_external_delta(g.handle, None)
        g.close()
        hip.hipFree(bufs[r])
    o.close()
print("bad (without mask flips):", bad)
