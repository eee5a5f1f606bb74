import ctypes as C
import recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
text = sc.synthetic_text(30000)
g = sc.AmdBatchedSet(amd, input_size=42, hidden_size=1024, output_size=42, S=256, D=20, learn_rate=1e-5, seed=1)
g.load_text(text)
for i in range(40):
    g.char_step(text, i)
amd.rnn_amd_synchronize()
amd.rnn_amd_debug_chain_only.restype = C.c_float
amd.rnn_amd_debug_chain_only.argtypes = [rc.NetP, C.c_int, C.c_int]
for reps in (5, 50):
    print("chain alone: %.2f us per launch over %d x 20 launches" % (amd.rnn_amd_debug_chain_only(g.net, 256, reps), reps))
