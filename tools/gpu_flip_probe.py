"""Development probe: one text-step shape generation by generation against the oracle; at the first generation whose
hidden zero masks differ, print the units and their values on both sides (a value within rounding of zero on the side
that has it is a rounding flip): gpu_flip_probe.py hidden alphabet S D act method noise learn_rate seed [steps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import recur_ctypes as rc
import scenarios as sc

a = sys.argv[1:]
hidden, A, S, D, act, method = (int(x) for x in a[:6])
noise, lr, seed = float(a[6]), float(a[7]), int(a[8])
steps = int(a[9]) if len(a) > 9 else D + 4
amd = rc.load_amd()
kw = dict(input_size=A, hidden_size=hidden, output_size=A, S=S, D=D, learn_rate=lr, seed=seed, activation=act, noise=noise)
if os.environ.get("VARIANCE"):  # large initial weights: RECLIP20 units reach 20
    kw["variance"] = float(os.environ["VARIANCE"])
text = sc.synthetic_text(int(os.environ.get("TEXT_LEN", "8000")), alphabet=A)
g, o = sc.AmdBatchedSet(amd, **kw), sc.OracleSet(**kw)
for i in range(steps):
    g.char_step(text, i, method, 0.9)
    o.char_step(text, i, method, 0.9)
    sg, so = g.snapshot(), o.snapshot()
    hg, ho = sg["hidden"], so["hidden"]
    d = np.argwhere((hg != 0) != (ho != 0))
    nh = int(((sg["hist"] != 0) != (so["hist"] != 0)).sum())
    dd = np.abs(sg["ih_delta"] - so["ih_delta"])
    print("   history flips %d, ih_delta rel %.2e (max element %.2e of max)" % (
        nh, np.linalg.norm(dd) / max(np.linalg.norm(so["ih_delta"]), 1e-30), dd.max() / max(np.abs(so["ih_delta"]).max(), 1e-30)))
    rel = np.abs(hg - ho).max() / max(np.abs(ho).max(), 1e-30)
    print("   units at RECLIP20's ceiling: %d / %d, ceiling flips %d" % (int((ho >= 20).sum()), int((hg >= 20).sum()), int(((hg >= 20) != (ho >= 20)).sum())))
    print("generation %d: %d flips, hidden max|diff|/max %.2e, max|hidden| %.3g, ih_w diff %.2e" % (
        i, len(d), rel, np.abs(ho).max(), np.abs(sg["ih_w"] - so["ih_w"]).max() / np.abs(so["ih_w"]).max()), flush=True)
    if len(d):
        for s_, y in d[:10]:
            print("   stream %d unit %d: product %.9g oracle %.9g" % (s_, y, hg[s_, y], ho[s_, y]))
        if not os.environ.get("GO_ON"):
            break
g.close()
o.close()
