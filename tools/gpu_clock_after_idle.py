"""Development probe (GPU box, build with tools/mkabl.sh stamps -DPC_STAMPS): the shader clock inside the chain launch (s_memtime
against s_memrealtime between barrier 0 and the tail, workgroup 0) of the n-th generation after the device has been idle --
how fast the clock comes back after a synchronisation, i.e. what a short timed window (bench.py --steps 20) sees."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
text = sc.synthetic_text(60000)
g = sc.AmdBatchedSet(amd, input_size=42, hidden_size=1024, output_size=42, S=256, D=20, learn_rate=1e-5, seed=1)
g.load_text(text)
i = 0
for _ in range(300):
    amd.rnn_amd_set_char_step(g.handle, i, rc.WEIGHTED, 0.95); i += 1
amd.rnn_amd_synchronize()
for idle_ms in (0.0, 0.2, 2.0, 20.0):
    for n in (1, 2, 3, 5, 10, 20, 50, 200):
        for _ in range(300):
            amd.rnn_amd_set_char_step(g.handle, i, rc.WEIGHTED, 0.95); i += 1
        amd.rnn_amd_synchronize()
        if idle_ms: time.sleep(idle_ms / 1000.0)
        t0 = time.perf_counter()
        for _ in range(n):
            amd.rnn_amd_set_char_step(g.handle, i, rc.WEIGHTED, 0.95); i += 1
        amd.rnn_amd_synchronize()
        el = time.perf_counter() - t0
        tb = np.zeros(32, np.uint64)
        amd.ramd_chain_tail_stamps(C.c_void_p(tb.ctypes.data))
        rt, ct = int(tb[26]) - int(tb[24]), int(tb[27]) - int(tb[25])
        print("idle %5.1f ms, generation %3d after it: chain loop %.2f us at %.3f GHz; the %d generations %.1f us each by the host's clock" % (
            idle_ms, n, rt / 100.0, ct / (rt * 10.0), n, el / n * 1e6))
