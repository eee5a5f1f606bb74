"""Development probe: generation rate of the batched text step at the BASELINE.json config shapes."""
import sys
import time
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
text = sc.synthetic_text(30000)
for name, H, S, D in (("configs[1] text 1024/64/20", 1024, 64, 20), ("configs[2] classify-like 512/128/30", 512, 128, 30),
                      ("north star 1024/256/20", 1024, 256, 20), ("configs[4] rnnca-like 2048/512/10", 2048, 512, 10),
                      ("1024/512/20", 1024, 512, 20)):
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    g = sc.AmdBatchedSet(amd, input_size=42, hidden_size=H, output_size=42, S=S, D=D, learn_rate=1e-5, seed=1)
    g.load_text(text)
    for i in range(D + 6):
        g.char_step(text, i)
    amd.rnn_amd_synchronize()
    n = 60
    t0 = time.time()
    for i in range(D + 6, D + 6 + n):
        g.char_step(text, i)
    amd.rnn_amd_synchronize()
    dt = time.time() - t0
    flops = S * (2 * g.I * g.H + 3 * 2 * g.H * g.O + D * 4 * g.I * g.H)
    print("shape %-36s %9.0f stream-timesteps/s  %7.1f us/generation  %5.1f TFLOP/s" %
          (name, n * S / dt, 1e6 * dt / n, flops * n / dt / 1e12))
    g.close()
