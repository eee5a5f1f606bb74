"""The oracle's restatements of the CALLERS' losses (oracle/recur_oracle.c: orc_multi_softmax_error,
orc_multitext_train, orc_grouped_softmax_error, orc_sigmoid_mse_error), pinned to what the reference's
own primitives give when walked through the callers' control flow on the reference's own net
(tests/golden/ref_callers.npz, written by tests/golden/make_golden_callers.py in the container that
has /root/reference).  CPU only.  References: charmodel-multi-predict.c:19-58, 234-281;
gstclassify.c:2070-2127; gstrnnca.c:693-716."""
import ctypes as C
import os

import numpy as np
import pytest

import callers_cases as cc
import recur_ctypes as rc
import scenarios as sc

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_callers.npz"))
RTOL = 2e-6  # the forward and backward passes under these losses are pinned at this level (test_oracle_golden.py)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("name", sorted(cc.MULTI_CASES))
def test_multi_head_trainer_matches_the_reference_walk(name):
    """Every symbol's own-head error and merged error ranges, the generator (the leakage draws and the
    noise draws are its only users), and the net after all passes."""
    c = cc.MULTI_CASES[name]
    kw = cc.multi_kwargs(c)
    o = sc.OracleSet(**kw)
    if c["method"] == rc.ADAGRAD:
        o.arrays()["ih_m"][:] = c["ballast"]  # ADAGRAD accumulates in the momentum arrays (recur-nn.c:617-621)
        o.arrays()["ho_m"][:] = c["ballast"]
    text = cc.multi_text(c)
    # step by step through orc_multi_softmax_error first (a second oracle net), so that the per-symbol
    # values are compared as well as the end state
    p = sc.OracleSet(**kw)
    if c["method"] == rc.ADAGRAD:
        p.arrays()["ih_m"][:] = c["ballast"]
        p.arrays()["ho_m"][:] = c["ballast"]
    width = 2 * (c["classes"] + 1)
    ranges = (C.c_int * width)()
    errs, rows = [], []
    sums = []
    for cls, lo, hi in cc.multi_passes(c, len(text)):
        t = np.ascontiguousarray(text[lo:hi])
        e, h = C.c_float(0), C.c_float(0)
        o.orc.orc_multitext_train(o.z, 0, rc.u8ptr(t), len(t), c["A"], cls, c["leakage"], c["method"], 0.9, c["batch"],
                                  C.byref(e), C.byref(h))
        sums.append([e.value, h.value])
        # the same pass, loss call by loss call
        countdown = c["batch"] - int(p.arrays()["generation"][0]) % c["batch"]
        for i in range(len(t) - 1):
            p.orc.orc_advance(p.z, 0)
            err = p.orc.orc_multi_softmax_error(p.z, 0, int(t[i]), int(t[i + 1]), cls, c["A"], c["leakage"], ranges)
            got = list(ranges)
            end = got.index(-1)
            rows.append(got[:end] + [-1] * (width - end))
            errs.append(err)
            if countdown == 0:
                p.orc.orc_apply_learning(p.z, c["method"], 0.9)
                countdown = c["batch"]
                p.orc.orc_calc_deltas(p.z, 0, 0, ranges)
            else:
                p.orc.orc_calc_deltas(p.z, 0, 1, ranges)
            countdown -= 1
    assert np.isfinite(GOLD[name + ".err"]).all() and np.isfinite(GOLD[name + ".ih_w"]).all()
    assert np.array_equal(np.array(rows, np.int32), GOLD[name + ".ranges"])
    assert np.allclose(np.array(errs, np.float32), GOLD[name + ".err"], rtol=1e-5, atol=1e-6)
    assert np.allclose(np.array(sums, np.float32), GOLD[name + ".sums"], rtol=1e-5)
    for who in (o, p):
        snap = who.snapshot()
        assert np.array_equal(snap["rng"], GOLD[name + ".rng"])
        assert np.array_equal(snap["generation"], GOLD[name + ".generation"])
        assert np.array_equal(snap["index"], GOLD[name + ".index"])
        for k in ("ih_w", "ho_w", "ih_m", "ho_m", "ih_delta", "ho_delta", "hidden", "output", "o_error", "hist"):
            assert rel(snap[k], GOLD["%s.%s" % (name, k)]) < RTOL, (k, rel(snap[k], GOLD["%s.%s" % (name, k)]))
    # the trained net's per-head cross entropy (orc_multi_cross_entropy against the reference walk)
    probe = np.ascontiguousarray(text[:60])
    ent = (C.c_double * c["classes"])()
    o.orc.orc_multi_cross_entropy(o.z, 0, rc.u8ptr(probe), len(probe), c["A"], ent, 5)
    assert np.allclose(np.array(list(ent)), GOLD[name + ".xent"], rtol=1e-5)
    o.close()
    p.close()


@pytest.mark.parametrize("name", sorted(cc.GROUP_CASES))
def test_grouped_softmax_error_matches_the_reference_walk(name):
    c = cc.GROUP_CASES[name]
    answers, targets, weight = cc.group_inputs(c)
    assert np.array_equal(answers, GOLD[name + ".answers"]) and np.array_equal(targets, GOLD[name + ".targets"])
    S, O = answers.shape
    o = sc.OracleSet(input_size=4, hidden_size=8, output_size=O, S=S, D=2)
    sizes = np.array(c["sizes"], np.int32)
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32)
    a = o.arrays()
    a["output"][:, :O] = answers
    for s in range(S):
        wins, wrong = C.c_int(0), C.c_float(0)
        tg = np.ascontiguousarray(targets[s])
        n = o.orc.orc_grouped_softmax_error(o.z, s, len(sizes), rc.iptr(offs), rc.iptr(sizes), rc.iptr(tg),
                                            rc.fptr(weight) if weight is not None else None, C.byref(wins),
                                            C.byref(wrong))
        want = GOLD[name + ".stats"][s]
        assert (n, wins.value) == (int(want[0]), int(want[1]))
        assert abs(wrong.value - want[2]) <= 1e-5 * max(1.0, abs(want[2]))
    got = o.arrays()["o_error"][:, :O]
    want = GOLD[name + ".errors"]
    trained = GOLD[name + ".stats"][:, 0] > 0
    # (a stream with no trained group keeps whatever its untouched groups held: the reference only zeroes
    # the groups it visits, and so does the restatement -- both started from zeros here)
    assert np.allclose(got[trained], want[trained], rtol=1e-5, atol=1e-7)
    assert not got[~trained].any()
    o.close()


def test_sigmoid_slope_error_matches_the_reference_walk():
    answers, targets = cc.sigmoid_inputs()
    assert np.array_equal(answers, GOLD["sigmoid.answers"])
    S = len(answers)
    o = sc.OracleSet(input_size=4, hidden_size=8, output_size=3, S=S, D=2)
    o.arrays()["output"][:, :3] = answers
    for s in range(S):
        o.orc.orc_sigmoid_mse_error(o.z, s, rc.fptr(np.ascontiguousarray(targets[s])), 3)
    a = o.arrays()
    assert np.allclose(a["output"][:, :3], GOLD["sigmoid.activated"], rtol=1e-6, atol=1e-30)  # sigmoid IN PLACE
    assert np.allclose(a["o_error"][:, :3], GOLD["sigmoid.errors"], rtol=1e-5, atol=1e-9)
    o.close()
