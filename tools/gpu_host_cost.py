"""Development probe: host time to enqueue one generation vs device time per generation."""
import time
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
text = sc.synthetic_text(30000)
g = sc.AmdBatchedSet(amd, input_size=42, hidden_size=1024, output_size=42, S=256, D=20, learn_rate=1e-5, seed=1)
g.load_text(text)
for i in range(40):
    g.char_step(text, i)
amd.rnn_amd_synchronize()
for n in (5, 20, 60):
    amd.rnn_amd_synchronize()
    t0 = time.perf_counter()
    for i in range(40, 40 + n):
        amd.rnn_amd_set_char_step(g.handle, i, rc.WEIGHTED, 0.95)
    t1 = time.perf_counter()
    amd.rnn_amd_synchronize()
    t2 = time.perf_counter()
    print("host %d generations: enqueue %.1f us/gen, total %.1f us/gen" % (n, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
g.close()
