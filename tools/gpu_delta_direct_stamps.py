"""Development probe (GPU box, build with tools/mkabl.sh stamps -DPC_STAMPS): where k_delta_direct's time goes, from
s_memrealtime stamps (10 ns ticks) of thread 0 of workgroups 0, 64, 128 and 192 of the LAST launch."""
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
buf = np.zeros((8, 8), np.uint64)
amd.ramd_ddir_stamps(C.c_void_p(buf.ctypes.data))
names = ["start", "ring requested, hook done", "flags there", "loop done", "tile in LDS", "barrier passed", "end"]
t0 = int(buf[:, 0].min())
for w in range(4):
    print("workgroup %3d: " % (64 * w) + "  ".join("%s %.2f" % (n, (int(buf[w, i]) - t0) / 100.0) for i, n in enumerate(names)))
for w in range(4):
    print("workgroup %3d, loop done per wave: " % (64 * w) + " ".join("%.1f" % ((int(x) - t0) / 100.0) for x in buf[4 + w]))

if hasattr(amd, "ramd_ddir_clocks"):
    ck = np.zeros((4, 4), np.uint64)
    amd.ramd_ddir_clocks(C.c_void_p(ck.ctypes.data))
    for w in range(4):
        rt, ct = int(ck[w, 2]) - int(ck[w, 0]), int(ck[w, 3]) - int(ck[w, 1])
        if rt > 0:
            print("workgroup %3d, the loop (wave 0) by both clocks: %.2f us, %d shader clocks: %.3f GHz" % (64 * w, rt / 100.0, ct, ct / (rt * 10.0)))
