"""Development probe: plain run, a run inside a one-rank RCCL group, a plain run again -- must the last equal the first?"""
import ctypes as C, sys, json, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
amd.rnn_amd_use_device(0, None)
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
plain = run(10)
plain2 = run(10)
buf = C.create_string_buffer(128)
assert amd.rnn_amd_dist_get_id(buf) == 0
assert amd.rnn_amd_dist_init(0, 1, buf) == 0
joined = run(10)
amd.rnn_amd_dist_finalize()
after = run(10)
after2 = run(10)
for k in ("ih_w", "ho_w", "ih_m", "ih_delta", "ho_delta", "hidden", "hist"):
    print(k, rc.rel_err(plain2[k], plain[k]), rc.rel_err(joined[k], plain[k]), rc.rel_err(after[k], plain[k]), rc.rel_err(after2[k], after[k]))
