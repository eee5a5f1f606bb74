"""Development probe: does the same run repeat bit for bit within a process (mapping modes, side streams)?"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
text = sc.synthetic_text(6000)
kw = dict(input_size=42, hidden_size=1024, output_size=42, S=64, D=5, learn_rate=1e-4, seed=21)
def run(n):
    g = sc.AmdBatchedSet(amd, **kw)
    g.load_text(text)
    for i in range(n):
        g.char_step(text, i, rc.WEIGHTED, 0.9)
    s = g.snapshot()
    g.close()
    return s
a = run(10)
b = run(10)
if len(sys.argv) > 1:
    amd.ramd_note_side_stream()
c = run(10)
for k in ("ih_w", "ho_w", "ih_delta", "ho_delta", "hidden"):
    print(k, rc.rel_err(b[k], a[k]), rc.rel_err(c[k], a[k]))
