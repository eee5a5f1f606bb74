"""Development probe: run N generations of the north-star workload (for profilers)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
text = sc.synthetic_text(30000)
S = int(os.environ.get("TUNE_S", "256")); H = int(os.environ.get("TUNE_H", "1024")); N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = sc.AmdBatchedSet(amd, input_size=42, hidden_size=H, output_size=42, S=S, D=int(os.environ.get("TUNE_D", "20")), learn_rate=1e-5, seed=1)
g.load_text(text)
for i in range(N):
    g.char_step(text, i)
amd.rnn_amd_synchronize()
print("done")
