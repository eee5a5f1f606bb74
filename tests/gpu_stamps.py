import ctypes as C
import numpy as np
import recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
text = sc.synthetic_text(30000)
g = sc.AmdBatchedSet(amd, input_size=42, hidden_size=1024, output_size=42, S=256, D=20, learn_rate=1e-5, seed=1)
g.load_text(text)
for i in range(40):
    g.char_step(text, i)
amd.rnn_amd_synchronize()
buf = np.zeros(20 * 4 * 64, np.uint64)
amd.rnn_amd_debug_read_slab.argtypes = [rc.NetP, C.c_size_t, C.c_void_p, C.c_size_t]
amd.rnn_amd_debug_read_slab(g.net, 12 * g.I * g.H, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
b = buf.reshape(20, 4, 64).astype(np.int64)
for blk, base in ((0, 0), (100, 2)):
    for t in (5, 12):
        c, l = b[t, base], b[t, base + 1]
        t0 = c[0]
        print("block %d step %d (10 ns ticks from the compute wave's start)" % (blk, t))
        print("  compute: at-barrier ", [int(c[1 + 2 * s] - t0) for s in range(8)])
        print("  compute: past-barrier", [int(c[2 + 2 * s] - t0) for s in range(7)])
        print("  loader : landed      ", [int(l[1 + 2 * s] - t0) for s in range(8)])
        print("  loader : past-barrier", [int(l[2 + 2 * s] - t0) for s in range(8)])
        print("  loop done %d, end %d" % (c[40] - t0, c[41] - t0))
