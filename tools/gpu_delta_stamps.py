import ctypes as C, os, sys
sys.path.insert(0, "tests")
import numpy as np
import recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
text = sc.synthetic_text(30000)
g = sc.AmdBatchedSet(amd, input_size=42, hidden_size=1024, output_size=42, S=256, D=20, learn_rate=1e-5, seed=1)
g.load_text(text)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    amd.rnn_amd_set_char_step(g.handle, i, rc.WEIGHTED, 0.95)
buf = np.zeros((1, 64, 8), np.uint64)
amd.ramd_delta_stamps(C.c_void_p(buf.ctypes.data))
t0 = int(buf[0, 0, 4])
us = lambda x: (int(x) - t0) / 100.0
print("start->first barrier release: see stage 1; end of loop at %.2f us" % us(buf[0,0,6]))
print("shader clock over the loop: %.0f MHz" % ((int(buf[0, 1, 7]) - int(buf[0, 0, 7])) / us(buf[0, 0, 6])))
print("stages that took the all-ones path: %d of 40" % int(buf[0, :40, 3].sum()))
prev = None
for st in range(1, 40):
    a, b = us(buf[0, st, 4]), us(buf[0, st, 5])
    print("stage %2d: arrive %7.2f release %7.2f wait %5.2f  since prev release %5.2f" % (st, a, b, b - a, (b - prev) if prev else 0))
    prev = b
