"""Development probe: hidden rows after a few text steps, fused forward against the separate kernels."""
import os, sys, subprocess, numpy as np
if len(sys.argv) > 1:
    import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import recur_ctypes as rc, scenarios as sc
    amd = rc.load_amd()
    H, S, D, steps = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    g = sc.AmdBatchedSet(amd, input_size=42, hidden_size=H, output_size=42, S=S, D=D, learn_rate=1e-5, seed=3)
    text = sc.synthetic_text(30000)
    for i in range(steps):
        g.char_step(text, i, rc.WEIGHTED, 0.95)
    s = g.snapshot()
    np.savez(sys.argv[1], hidden=s["hidden"], hist=s["hist"], index=s["index"])
    sys.exit(0)
for H, S, D, steps in ((512, 64, 8, 3), (256, 32, 4, 3), (128, 32, 4, 3), (1024, 128, 20, 3), (768, 64, 8, 3), (1536, 64, 6, 3), (64, 32, 4, 3), (96, 40, 5, 3), (2016, 32, 3, 3)):
    for name, env in (("a", {}), ("b", {"RECUR_AMD_NO_FWD_FUSED": "1"})):
        subprocess.check_call([sys.executable, __file__, "/tmp/fd_%s.npz" % name, str(H), str(S), str(D), str(steps)],
                              env=dict(os.environ, **env), stderr=subprocess.DEVNULL)
    a, b = np.load("/tmp/fd_a.npz"), np.load("/tmp/fd_b.npz")
    for k in ("hidden",):
        d = np.abs(a[k].astype(np.float64) - b[k])
        bad = np.argwhere(d > 1e-5 * (1 + np.abs(b[k])))
        print(H, S, D, steps, k, "max diff %.3g" % d.max(), "bad", len(bad), bad[:6].tolist())
