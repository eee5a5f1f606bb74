"""Live cross-check of the oracle against the real reference library
(oracle/_ref/librecur_ref.so).  Skipped where that file is absent; the committed
golden vectors (test_oracle_golden.py) are the portable form of the same pin."""
import numpy as np
import pytest

import recur_ctypes as rc
import scenarios as sc

pytestmark = pytest.mark.skipif(not rc.have_ref(), reason="oracle/_ref not built here")


def test_abi_struct_sizes_match_reference():
    import ctypes as C
    ref = rc.load_ref()
    assert ref.ref_sizeof_net() == C.sizeof(rc.RecurNN)
    assert ref.ref_sizeof_bptt() == C.sizeof(rc.RecurNNBPTT)


@pytest.mark.parametrize("act", [rc.RELU, rc.RESQRT, rc.RECLIP20])
@pytest.mark.parametrize("method", [rc.WEIGHTED, rc.NESTEROV])
def test_oracle_tracks_reference_over_a_run(act, method):
    ref = rc.load_ref()
    text = sc.synthetic_text(3000)
    kw = dict(input_size=42, hidden_size=57, output_size=42, S=5, D=9, activation=act, learn_rate=5e-3,
              seed=17)
    a = sc.ApiSet(ref, softmax_best_guess=ref.ref_softmax_best_guess, **kw)
    o = sc.OracleSet(**kw)
    assert sc.compare(o.snapshot(), a.snapshot(), 0) == []  # initialisation is bit-exact
    for i in range(25):
        a.char_step(text, i, method, 0.9)
        o.char_step(text, i, method, 0.9)
    assert sc.compare(o.snapshot(), a.snapshot(), 2e-6) == []
    a.close()
    o.close()
