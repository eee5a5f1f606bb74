"""Development probe: sweeps the split-K factors of the GEMMs (env overrides read
at every launch) so that a rocprofv3 kernel trace can rank them by grid size."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
text = sc.synthetic_text(30000)
S = int(os.environ.get("TUNE_S", "256"))
g = sc.AmdBatchedSet(amd, input_size=42, hidden_size=1024, output_size=42, S=S, D=20, learn_rate=1e-5, seed=1)
g.load_text(text)
i = 0
for _ in range(25):
    g.char_step(text, i); i += 1
amd.rnn_amd_synchronize()
for name in sys.argv[1:]:
    var, vals = name.split("=")
    for v in vals.split(","):
        os.environ[var] = v
        for _ in range(4):
            g.char_step(text, i); i += 1
        amd.rnn_amd_synchronize()
    del os.environ[var]
print("done")
