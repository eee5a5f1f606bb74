"""Development probe: training sets far larger than the north star's 256 streams (DESIGN.md section 2 sizes the
layout for thousands of streams per GPU; the fixed GPU tests stop at 640).
  part A, parity: one generation of the batched text step from the device's own warmed-up state against the oracle
          (hidden 256 / 4096 streams / depth 6 and hidden 1024 / 2048 streams / depth 4: the one-launch chain in
          several windows, the LDS-DMA delta GEMM over K = S * D rows), 1e-4 relative and max-element;
  part B, rate: hidden 1024 / depth 20 at 1024, 4096 and 16384 streams.
    gpu_large_sets.py [parity|rate]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import recur_ctypes as rc
import replay
import scenarios as sc

amd = rc.load_amd()
what = sys.argv[1] if len(sys.argv) > 1 else "all"
text = sc.synthetic_text(200000)
bad = 0

if what in ("all", "parity"):
    for kw in (dict(input_size=42, hidden_size=256, output_size=42, S=4096, D=6, learn_rate=1e-5, seed=3),
               dict(input_size=42, hidden_size=1024, output_size=42, S=2048, D=4, learn_rate=1e-5, seed=4),
               dict(input_size=42, hidden_size=1024, output_size=42, S=2048, D=4, learn_rate=1e-5, seed=5),
               dict(input_size=42, hidden_size=1024, output_size=42, S=1536, D=4, learn_rate=1e-5, seed=6)):
        g = sc.AmdBatchedSet(amd, **kw)
        g.load_text(text)
        warm = kw["D"] + 3
        for i in range(warm):
            g.char_step(text, i, rc.WEIGHTED, 0.95)
        snap = g.snapshot()
        o = sc.OracleSet(**kw)
        a = o.arrays()
        for k in ("ih_w", "ho_w", "ih_m", "ho_m", "hist", "hidden", "index", "min_error_factor"):
            a[k][:] = snap[k]
        a["generation"][:] = snap["generation"]
        t0 = time.time()
        g.char_step(text, warm, rc.WEIGHTED, 0.95)
        o.char_step(text, warm, rc.WEIGHTED, 0.95)
        sg, so = g.snapshot(), o.snapshot()
        msgs = sc.compare(sg, so, 1e-4, keys=["ih_delta", "ho_delta", "ih_w", "ho_w", "ih_m", "ho_m", "hidden", "output",
                                              "o_error", "hist", "min_error_factor", "ih_scale"],
                          exact=("index", "generation"))
        same_mask = bool(np.array_equal(sg["hidden"] != 0, so["hidden"] != 0))
        print("parity hidden %d / %d streams / depth %d: %s, zero masks %s, mean depth %.2f, worst rel err %.2g (oracle %.0f s)"
              % (kw["hidden_size"], kw["S"], kw["D"], "ok" if not msgs else msgs, "equal" if same_mask else "DIFFER",
                 float(np.mean(so["bptt_depth"])),
                 max(rc.rel_err(sg[k], so[k]) for k in ("ih_delta", "ho_delta", "ih_w", "hidden")), time.time() - t0),
              flush=True)
        if not same_mask:
            # a pre-activation within rounding of zero lands on different sides in the two summation orders: the
            # entry is 0 on one side and tiny on the other (then the generation's deltas differ by that unit's row)
            d = (sg["hidden"] != 0) != (so["hidden"] != 0)
            print("   %d of %d hidden entries differ in the mask; largest |value| there: product %.3g, oracle %.3g"
                  % (int(d.sum()), d.size, float(np.abs(sg["hidden"][d]).max()), float(np.abs(so["hidden"][d]).max())))
            flips = int(d.sum())
        bad += (bool(msgs) and same_mask)
        o.close()
        g.close()

if what in ("all", "rate"):
    for S in (256, 1024, 4096, 16384):
        H, D = 1024, 20
        g = sc.AmdBatchedSet(amd, input_size=42, hidden_size=H, output_size=42, S=S, D=D, learn_rate=1e-5, seed=1)
        g.load_text(text)
        for i in range(D + 6):
            g.char_step(text, i)
        amd.rnn_amd_synchronize()
        n = max(4, 12800 // S)
        t0 = time.time()
        for i in range(D + 6, D + 6 + n):
            g.char_step(text, i)
        amd.rnn_amd_synchronize()
        dt = time.time() - t0
        flops = S * (2 * g.I * g.H + 3 * 2 * g.H * g.O + D * 4 * g.I * g.H)
        print("rate hidden 1024 / %5d streams / depth 20: %9.0f stream-timesteps/s  %9.1f us/generation  %5.1f TFLOP/s"
              % (S, n * S / dt, 1e6 * dt / n, flops * n / dt / 1e12), flush=True)
        g.close()
print("bad (without mask flips):", bad)
sys.exit(1 if bad else 0)
