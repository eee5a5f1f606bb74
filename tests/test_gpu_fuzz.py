"""Random sequences of the reference's API calls on the product and on the COMPILED REFERENCE
(oracle/_ref/librecur_ref.so, built from /root/reference by oracle/Makefile; it travels to the GPU box
like the product's own library), the same driver code on both, compared after every call:
tools/gpu_fuzz_api.py.  Per-net calls, batched set calls, host-side edits, forgotten histories, clones
made and deleted in between, weight noise, the fused single-net call with and without batching,
accumulation after rnn_bptt_clear_deltas, error ranges, dense inputs with an active mask, every
conditioning step, the optimiser families, bottom-layer nets, shapes up to hidden 512 / 70 streams.
Fixed seeds: the run is deterministic.  (In development the tool found the seven defects whose
regression tests sit in test_gpu_parity.py: split-K workspace of narrow nets under wide output layers,
the delta arrays after rnn_bptt_calculate, RECLIP20 units at the ceiling, the bottom layer's shared
input buffer and its error accumulator under a clipped stream, a division by zero for sub-range sets of tiny nets.)"""
import os
import re
import subprocess
import sys

import pytest

import recur_ctypes as rc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not rc.have_ref(), reason="oracle/_ref/librecur_ref.so was not built (needs /root/reference)")
@pytest.mark.parametrize("seed", [30, 31, 35])
def test_random_call_sequences_against_the_compiled_reference(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_fuzz_api.py"), str(seed), "10", "30"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    m = re.match(r"bad: (\d+), trials ended by a mask flip: (\d+)", last)
    assert m, r.stdout[-2000:]
    ok = len(re.findall(r"operations ok", r.stdout))
    assert int(m.group(1)) == 0, [l for l in r.stdout.splitlines() if "MISMATCH" in l][:3]
    assert ok + int(m.group(2)) == 10 and ok >= 8     # every trial ran to its end or to a rounding-level flip
