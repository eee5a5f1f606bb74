"""GPU probe: the one-launch BPTT chain beside ANOTHER PROCESS's kernels on the same GPU.  A second
process launches 256-workgroup kernels back to back (build/delta_direct_microbench: one workgroup per CU, 139 KB of LDS
each) while this one trains the north-star set; the dispatcher interleaves the two processes' workgroups, so a chain
workgroup's NUMBER says nothing about the XCD it lands on.  With the seat taken from the CU a workgroup runs on (the
default) the generations are correct and nothing gives up; with RECUR_AMD_XCD_TABLE=0 RECUR_AMD_XCD_STATIC=1 (round 4's
default) the first misplaced workgroup raises abort code 2.  Prints the parity of one generation from synchronised state
against the oracle, made while the other process is running, and the rate.
usage: python tools/gpu_cotenant_probe.py [generations=200]"""
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import recur_ctypes as rc
import scenarios as sc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
amd = rc.load_amd()
kw = dict(input_size=42, hidden_size=1024, output_size=42, S=256, D=20, learn_rate=1e-5, seed=1)
g = sc.AmdBatchedSet(amd, **kw)
text = sc.synthetic_text(30000)
g.load_text(text)
for i in range(24):
    g.char_step(text, i, rc.WEIGHTED, 0.95)
amd.rnn_amd_synchronize()


def _die_with_parent():
    # PR_SET_PDEATHSIG = 1, SIGKILL: this process may end by abort() (the workgroup-number map's give-up is what the
    # second half of the test provokes) -- the other process must not outlive it and slow down whatever runs next
    import ctypes
    ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, 9, 0, 0, 0)


other = subprocess.Popen([os.path.join(ROOT, "build", "delta_direct_microbench"), "4000"], stdout=subprocess.DEVNULL,
                         stderr=subprocess.DEVNULL, preexec_fn=_die_with_parent)
print("other process: pid %d" % other.pid, flush=True)
time.sleep(1.5)  # (its checks are over, its timing loops run)
t0 = time.perf_counter()
for i in range(24, 24 + N):
    g.char_step(text, i, rc.WEIGHTED, 0.95)
amd.rnn_amd_synchronize()
dt = time.perf_counter() - t0
alive = other.poll() is None
o = sc.OracleSet(**kw)
a = o.arrays()
for attempt in range(4):
    # one generation on both sides from the device's state; a generation in which a pre-activation within rounding of zero
    # takes its mask from the summation order is held to the flip bound (tests/test_gpu_parity.py:
    # test_mask_flips_stay_at_the_rounding_level_rate) and is not a parity case: take the next one
    snap = g.snapshot()
    for k in ("ih_w", "ho_w", "ih_m", "ho_m", "hist", "hidden", "index", "min_error_factor"):
        a[k][:] = snap[k]
    a["generation"][:] = snap["generation"]
    g.char_step(text, 24 + N + attempt, rc.WEIGHTED, 0.95)
    o.char_step(text, 24 + N + attempt, rc.WEIGHTED, 0.95)
    sg, so = g.snapshot(), o.snapshot()
    flipped = (sg["hidden"] != 0) != (so["hidden"] != 0)
    if not flipped.any():
        break
    print("generation %d: %d hidden values differ in being zero (largest %.1e): the next one" % (
        24 + N + attempt, flipped.sum(), np.abs(np.where(sg["hidden"][flipped] != 0, sg["hidden"][flipped], so["hidden"][flipped])).max()))
    assert 1e6 * flipped.sum() / flipped.size <= 10.0
    assert np.abs(np.where(sg["hidden"][flipped] != 0, sg["hidden"][flipped], so["hidden"][flipped])).max() < 1e-5
else:
    raise AssertionError("no generation without a rounding-level mask flip in 4 attempts")
still = other.poll() is None
errs = {k: max(rc.rel_err(sg[k], so[k]), rc.max_err(sg[k], so[k])) for k in ("ih_delta", "ho_delta", "ih_w", "ho_w", "hidden")}
print("errors by array: %s" % errs)
worst = max(errs.values())
print("%d generations beside the other process (%s at the end of the loop, %s at the compared generation): %.1f us per "
      "generation; one generation from synchronised state against the oracle: worst error %.2e" % (
          N, "running" if alive else "GONE", "running" if still else "gone", 1e6 * dt / N, worst))
other.kill()
other.wait()
assert worst <= 1e-4
print("ok")
