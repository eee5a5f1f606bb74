"""GPU probe (one GPU): what the exchange step costs A RANK, emulated -- VERDICT round 4, item 5c.  W virtual ranks =
W engines of this process, each a shard of S streams of the north-star net (hidden 1024, depth 20), driven in lock step
as a multi-GPU job drives its ranks: deltas on every rank, then the exchange + update on every rank.  Three forms:
  plain      every rank runs the one-call text step on its own shard (no exchange at all: the floor; the update is the
             weight-delta GEMM's epilogue)
  kernel     rnn_amd_set_char_step_deltas, then rnn_amd_set_apply_exchange: k_apply_xchg adds the W ranks' delta sums for
             its 1 / W of the weight arrays through peer pointers, updates there, stores the new weights into every
             rank's arrays (here all on one GPU: what is measured is the kernels' cost, not a link)
  unfused    rnn_amd_set_char_step_deltas, then rnn_apply_learning on every rank: the generation with the update as its
             own launch -- what the RCCL path runs around its all-reduce (whose own time over xGMI is not emulated)
Reported: microseconds per rank and generation (wall time of a lock-step round / W) and the difference to `plain`.
usage: python tools/gpu_exchange_emulated.py [W=8] [S=32] [generations=60]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import recur_ctypes as rc
import scenarios as sc

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 60
amd = rc.load_amd()
text = sc.synthetic_text(40000)
kw = dict(input_size=42, hidden_size=1024, output_size=42, D=20, learn_rate=1e-5, seed=1)


def make(exchange):
    ranks = [sc.AmdBatchedSet(amd, S=S, shard=(r * S, W * S), **kw) for r in range(W)]
    blobs = C.create_string_buffer(rc.RNN_AMD_EXCHANGE_BLOB_BYTES * W)
    for r, g in enumerate(ranks):
        g.load_text(text)
        amd.rnn_amd_set_shard(g.handle, r * S, W * S)
        if exchange:
            amd.rnn_amd_set_exchange_export(g.handle, C.byref(blobs, r * rc.RNN_AMD_EXCHANGE_BLOB_BYTES))
    if exchange:
        for r, g in enumerate(ranks):
            assert amd.rnn_amd_set_exchange_join(g.handle, r, W, blobs, None, 1) == 0
    return ranks


def run(form):
    ranks = make(form == "kernel")

    def generation(i):
        if form == "plain":
            for g in ranks:
                amd.rnn_amd_set_char_step(g.handle, i, rc.WEIGHTED, 0.95)
        else:
            for g in ranks:
                amd.rnn_amd_set_char_step_deltas(g.handle, i)
            for g in ranks:
                if form == "kernel":
                    amd.rnn_amd_set_apply_exchange(g.handle, rc.WEIGHTED, 0.95)
                else:
                    amd.rnn_apply_learning(g.net, rc.WEIGHTED, 0.95)
    for i in range(26):
        generation(i)
    amd.rnn_amd_synchronize()
    t0 = time.perf_counter()
    for i in range(26, 26 + N):
        generation(i)
    amd.rnn_amd_synchronize()
    dt = time.perf_counter() - t0
    for g in ranks:
        if form == "kernel":
            amd.rnn_amd_set_exchange_leave(g.handle)
        g.close()
    return 1e6 * dt / N / W


print("%d virtual ranks x %d streams, hidden 1024, depth 20, %d generations" % (W, S, N))
base = None
for form in ("plain", "unfused", "kernel"):
    us = run(form)
    base = us if base is None else base
    print("  %-8s %7.1f us per rank and generation   (%+.1f against plain)" % (form, us, us - base), flush=True)
