"""Development probe: the batched text step across stream counts, depths and hidden sizes (a markdown table)."""
import sys, time, os
sys.path.insert(0, "tests")
import recur_ctypes as rc, scenarios as sc
amd = rc.load_amd()
text = sc.synthetic_text(60000)
SHAPES = [(1024, 16, 20), (1024, 32, 20), (1024, 64, 20), (1024, 128, 20), (1024, 256, 20), (1024, 512, 20), (1024, 1024, 20), (1024, 2048, 20),
                (1024, 256, 10), (1024, 256, 40), (512, 256, 20), (512, 1024, 20), (256, 1024, 20), (2048, 256, 20), (2048, 1024, 20)]
if len(sys.argv) > 1 and sys.argv[1] == "small":  # the shapes the one-sub-chain variants serve
    SHAPES = [(1024, 32, 20), (1024, 64, 20), (512, 128, 30), (512, 256, 20), (256, 128, 20), (256, 1024, 20)]
for H, S, D in SHAPES:
    g = sc.AmdBatchedSet(amd, input_size=42, hidden_size=H, output_size=42, S=S, D=D, learn_rate=1e-5, seed=1)
    g.load_text(text)
    for i in range(D + 6): g.char_step(text, i)
    amd.rnn_amd_synchronize()
    n = 60 if S <= 512 else 30
    t0 = time.time()
    for i in range(D + 6, D + 6 + n): g.char_step(text, i)
    amd.rnn_amd_synchronize()
    dt = time.time() - t0
    flops = S * (2 * g.I * g.H + 3 * 2 * g.H * g.O + D * 4 * g.I * g.H)
    print("| %d / %d / %d | %.0f k | %.0f | %.1f |" % (H, S, D, n * S / dt / 1e3, 1e6 * dt / n, flops * n / dt / 1e12))
    g.close()
