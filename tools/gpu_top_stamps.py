"""Development probe (GPU box, build with tools/mkabl.sh stamps -DPC_STAMPS): where k_text_top's time goes, from
s_memrealtime stamps (10 ns ticks) of workgroup 0, thread 0."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
text = sc.synthetic_text(30000)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = sc.AmdBatchedSet(amd, input_size=42, hidden_size=1024, output_size=42, S=S, D=20, learn_rate=1e-5, seed=1)
g.load_text(text)
for i in range(60):
    amd.rnn_amd_set_char_step(g.handle, i, rc.WEIGHTED, 0.95)
buf = np.zeros(16, np.uint64)
amd.ramd_top_stamps(C.c_void_p(buf.ctypes.data))
names = ["start", "hidden row in LDS", "output layer: waves' sums in LDS", "softmax done", "backprop: barrier passed", "end", "backprop: row stored, wave's sum in LDS"]
t0 = int(buf[0])
for i, n in enumerate(names):
    print("%-22s %6.2f us" % (n, (int(buf[i]) - t0) / 100.0))
for i, n in zip(range(8, 13), ["softmax wave: zeros counted", "max / min reduced", "exponentials in LDS", "sum formed",
                               "error row and best guess"]):
    print("  %-28s %6.2f us" % (n, (int(buf[i]) - t0) / 100.0))
