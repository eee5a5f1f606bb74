"""Replays the golden scenarios (golden_cases.py / golden/make_golden.py) on a
driver and returns the same dict of arrays the generator stored."""
import ctypes as C
import os

import numpy as np

import golden_cases as gc
import recur_ctypes as rc
import scenarios as sc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz")
_golden = None
_text = None


def golden():
    global _golden
    if _golden is None:
        _golden = np.load(GOLDEN)
    return _golden


def text():
    global _text
    if _text is None:
        _text = gc.synthetic_text_np()
    return _text


def golden_case(name):
    z = golden()
    pre = name + "."
    return {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}


# ------------------------------------------------------------------ training --

def train_oracle(name):
    c = gc.TRAIN_CASES[name]
    o = sc.OracleSet(**gc.case_kwargs(c))
    a = o.arrays()

    def set_m(x):
        a["ih_m"][:] = x
        if "b_m" in a:
            a["b_m"][:] = x
        a["ho_m"][:] = x

    def set_aux(x):
        a["ih_aux"][:] = x
        a["ho_aux"][:] = x
    gc.prepare(o, c, set_m, set_aux)
    trace = []
    gc.run_scenario(o, c, text(), lambda i: trace.append(a["ih_scale"].copy()))
    snap = o.snapshot()
    snap["h_error"], snap["i_error"] = snap.pop("err_a"), snap.pop("err_b")
    snap["ih_scale_trace"] = np.array(trace, np.float32)
    o.close()
    return snap


def train_api(lib, name, batched, sbg=None):
    """librecur_amd.so (or the reference) through the per-net ABI, or the
    product through its batched entry points."""
    c = gc.TRAIN_CASES[name]
    kw = gc.case_kwargs(c)
    if batched:
        a = sc.AmdBatchedSet(lib, **kw)
    else:
        a = sc.ApiSet(lib, softmax_best_guess=sbg, **kw)
    gc.prepare(a, c, lambda x: lib.rnn_set_momentum_values(a.net, x),
               lambda x: lib.rnn_set_aux_values(a.net, x))
    gc.run_scenario(a, c, text())
    snap = a.snapshot()
    b0 = a.net.contents.bptt.contents
    if c.get("aux"):
        snap["ih_aux"] = rc.view(b0.ih_aux, a.I, a.H).copy()
        snap["ho_aux"] = rc.view(b0.ho_aux, a.H, a.O).copy()
    snap["h_error"] = np.stack([rc.view(a.nets[j].contents.bptt.contents.h_error, a.I).copy()
                                for j in range(a.S)])
    snap["i_error"] = np.stack([rc.view(a.nets[j].contents.bptt.contents.i_error, a.I).copy()
                                for j in range(a.S)])
    a.close()
    return snap


# --------------------------------------------------------------- conditioning --

COND_BITS = ((0, rc.COND_USE_SCALE), (2, rc.COND_USE_ZERO), (3, rc.COND_USE_LAWN_MOWER),
             (4, rc.COND_USE_TALL_POPPY), (6, rc.COND_USE_RAND))


def cond_api(lib, bit, flag):
    net = lib.rnn_new(10, 31, 5, rc.FLAG_OWN_BPTT | rc.FLAG_OWN_WEIGHTS | flag, 21, None, 4, 0.01, 0.9,
                      0.0, rc.RELU)
    lib.rnn_randomise_weights_auto(net)
    n = net.contents
    w = rc.view(n.ih_weights, n.ih_size)
    w[7] = 14.5
    w[11] = -12.0
    w[13] = 1e-36
    rc.view(n.bptt.contents.ih_momentum, n.ih_size)[5] = 1e-37
    if hasattr(lib, "rnn_amd_host_written"):
        lib.rnn_amd_host_written(net, rc.RNN_AMD_WEIGHTS | rc.RNN_AMD_MOMENTUMS)
    n.generation = 8 + bit
    lib.rnn_condition_net(net)
    if hasattr(lib, "rnn_amd_sync_host"):
        lib.rnn_amd_sync_host(net, rc.RNN_AMD_EVERYTHING)
    res = dict(ih_w=w.copy(), ho_w=rc.view(n.ho_weights, n.ho_size).copy(),
               ih_m=rc.view(n.bptt.contents.ih_momentum, n.ih_size).copy())
    lib.rnn_delete_net(net)
    return res


def cond_oracle(bit, flag):
    o = sc.OracleSet(input_size=10, hidden_size=31, output_size=5, S=1, D=4, learn_rate=0.01, seed=21,
                     shape=rc.DIST_UNIFORM, perforation=0.7,
                     flags=rc.FLAG_OWN_BPTT | rc.FLAG_OWN_WEIGHTS | flag)
    a = o.arrays()
    w = a["ih_w"].reshape(-1)
    w[7] = 14.5
    w[11] = -12.0
    w[13] = 1e-36
    a["ih_m"].reshape(-1)[5] = 1e-37
    a["generation"][0] = 8 + bit
    o.orc.orc_condition(o.z, rc.FLAG_OWN_BPTT | rc.FLAG_OWN_WEIGHTS | flag)
    res = dict(ih_w=w.copy(), ho_w=a["ho_w"].reshape(-1).copy(), ih_m=a["ih_m"].reshape(-1).copy())
    o.close()
    return res


# ---------------------------------------------------------------- fused path --

def fused_api(lib, batch, sbg):
    a = sc.ApiSet(lib, softmax_best_guess=sbg, input_size=42, hidden_size=39, output_size=42, S=1, D=8,
                  learn_rate=1e-2, seed=13)
    t = text()
    for i in range(24):
        lib.rnn_bptt_advance(a.net)
        a.net_error_bptt(0, int(t[i]), int(t[i + 1]))
        a.net.contents.bptt.contents.momentum = 0.9
        lib.rnn_bptt_calculate(a.net, batch)
    snap = a.snapshot()
    a.close()
    return snap


def fused_oracle(batch):
    o = sc.OracleSet(input_size=42, hidden_size=39, output_size=42, S=1, D=8, learn_rate=1e-2, seed=13)
    t = text()
    c = C.c_int(0)
    for i in range(24):
        o.orc.orc_advance(o.z, 0)
        o.orc.orc_net_error_bptt(o.z, 0, int(t[i]), int(t[i + 1]), C.byref(c))
        o.orc.orc_bptt_calculate(o.z, 0, batch, 0.9)
    snap = o.snapshot()
    o.close()
    return snap


# ---------------------------------------------------------------- sparse top --

SPARSE_KW = dict(input_size=20, hidden_size=39, output_size=24, S=3, D=6, learn_rate=1e-2, seed=14)


def _sparse_inputs():
    rs = np.random.default_rng(9)
    steps = []
    for _ in range(10):
        per = []
        for _ in range(3):
            x = (rs.standard_normal(20).astype(np.float32) * (rs.random(20) < 0.5)).astype(np.float32)
            e = np.zeros(24, np.float32)
            e[4:12] = rs.standard_normal(8) * 0.1
            e[16:20] = rs.standard_normal(4) * 0.1
            per.append((x, e))
        steps.append(per)
    return steps


def sparse_api(lib, batched=False):
    ranges = (rc.ErrorRange * 3)((4, 8), (16, 4), (-1, 0))
    if batched:
        a = sc.AmdBatchedSet(lib, **SPARSE_KW)
    else:
        a = sc.ApiSet(lib, **SPARSE_KW)
    for per in _sparse_inputs():
        if batched:
            X = np.ascontiguousarray(np.stack([p[0] for p in per]))
            E = np.ascontiguousarray(np.stack([p[1] for p in per]))
            lib.rnn_amd_set_advance(a.handle)
            lib.rnn_amd_set_opinion(a.handle, rc.fptr(X), X.shape[1], None)
            lib.rnn_amd_set_put_o_error(a.handle, rc.fptr(E), E.shape[1])
            lib.rnn_amd_set_calc_deltas(a.handle, 0, ranges, None)
        else:
            for j, (x, e) in enumerate(per):
                lib.rnn_bptt_advance(a.nets[j])
                lib.rnn_opinion(a.nets[j], rc.fptr(x), 0.0)
                rc.view(a.nets[j].contents.bptt.contents.o_error, a.O)[:] = e
                lib.rnn_bptt_calc_deltas(a.nets[j], 1 if j else 0, ranges)
        lib.rnn_apply_learning(a.net, rc.WEIGHTED, 0.9)
    snap = a.snapshot()
    snap["h_error"] = np.stack([rc.view(a.nets[j].contents.bptt.contents.h_error, a.I).copy()
                                for j in range(3)])
    a.close()
    return snap


def sparse_oracle():
    o = sc.OracleSet(**SPARSE_KW)
    ranges = np.array([4, 8, 16, 4, -1, 0], np.int32)
    a = o.arrays()
    for per in _sparse_inputs():
        for j, (x, e) in enumerate(per):
            o.orc.orc_advance(o.z, j)
            o.orc.orc_opinion(o.z, j, rc.fptr(x), 0.0)
            a["o_error"][j, :] = e
            o.orc.orc_calc_deltas(o.z, j, 1 if j else 0, rc.iptr(ranges))
        o.orc.orc_apply_learning(o.z, rc.WEIGHTED, 0.9)
    snap = o.snapshot()
    snap["h_error"] = snap.pop("err_a")
    o.close()
    return snap


def check(got, want, rtol, keys=None, exact=("index", "generation"), elementwise=True, elem_floor=1e-2):
    """Assert helper: per array, the relative 2-norm error, the largest element error against the largest reference
    magnitude (one bad row cannot hide in a large array) AND -- north_star's "1e-4 relative" read element by element --
    |a - b| <= rtol |b| on every element within two orders of magnitude of the array's largest (rc.elem_err: the floor
    is where the reference's own two builds still agree to 3e-5; elements that cancelled further down differ by more
    than 1e-4 of themselves between ANY two summation orders).  elem_floor: 1e-2, or 1e-1 in the hot regime (streams
    soft-clipped, the error growing through the chain's steps), where the reference's own builds are 8.9e-5 apart at 1e-2
    and 4e-5 at 1e-1: profiles/r05_reference_elementwise_self_difference.txt."""
    keys = keys or [k for k in want if k in got and want[k].dtype.kind == "f"]
    bad = []
    for k in keys:
        e = rc.rel_err(got[k], want[k])
        if not e <= rtol:
            bad.append("%s rel err %.3g" % (k, e))
        m = rc.max_err(got[k], want[k])
        if not m <= rtol:
            bad.append("%s max element err %.3g of max|ref|" % (k, m))
        if elementwise and np.asarray(want[k]).dtype.kind == "f":
            el = rc.elem_err(got[k], want[k], elem_floor)
            if not el <= rtol:
                bad.append("%s element-wise err %.3g (elements >= %g of the largest)" % (k, el, elem_floor))
    for k in exact:
        if k in got and k in want and not np.array_equal(got[k], want[k]):
            bad.append("%s differs" % k)
    assert not bad, "; ".join(bad)
