"""Development probe (GPU box, build with tools/mkabl.sh stamps -DPC_STAMPS): where a half-step of
the one-launch chain spends its time, from s_memrealtime stamps (10 ns ticks) of workgroup
(XCD 0, seat 0): multiplying wave 4 and fetching wave 0."""
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
for i in range(30):
    amd.rnn_amd_set_char_step(g.handle, i, rc.WEIGHTED, 0.95)
buf = np.zeros((2, 64, 8), np.uint64)
amd.ramd_chain_stamps(C.c_void_p(buf.ctypes.data))
t0 = int(buf[1, 0, 0])
us = lambda x: (int(x) - t0) / 100.0 if int(x) else float("nan")
print("k | multiply: start finished-prev published dumped | fetch: start polled issued landed")
for k in range(41):
    c, l = buf[0, k], buf[1, k]
    print("%2d | M %7.2f fin %7.2f pub %7.2f end %7.2f | F %7.2f polled %7.2f issued %7.2f landed %7.2f" % (
        k, us(c[0]), us(c[2]), us(c[3]), us(c[1]), us(l[0]), us(l[2]), us(l[3]), us(l[4])))

if hasattr(amd, "ramd_chain_tail_stamps"):
    tb = np.zeros(32, np.uint64)
    amd.ramd_chain_tail_stamps(C.c_void_p(tb.ctypes.data))
    b0 = int(tb[0])
    names = ["kernel start", "tail start (wave 0)", "last flags seen (last wave)", "tail set up", "weight rows requested",
             "error rows requested", "items done (wave 0)", "barrier", "control done", "stream done", "tail start (wave 4)",
             "prologue: wave 4 set up", "prologue: panel requested", "prologue: panel landed", "prologue: barrier 0 passed (wave 4)",
             "prologue: first operands landed (wave 0)", "prologue: launch arguments read (wave 4)", "prologue: seat read (wave 4)",
             "prologue: view read (wave 4)", "prologue: before the tail's early requests (wave 4)", "prologue: after them (wave 4)"]
    print("tail of workgroup 0 (us from its first instruction):")
    if int(tb[24]) and int(tb[26]):
        rt, ct = int(tb[26]) - int(tb[24]), int(tb[27]) - int(tb[25])
        print("  the loop (barrier 0 -> tail, wave 4): %.2f us, %d shader clocks: %.3f GHz" % (rt / 100.0, ct, ct / (rt * 10.0)))
    for i, n in enumerate(names):
        print("  %-24s %8.2f" % (n, (int(tb[i]) - b0) / 100.0 if int(tb[i]) else float("nan")))
