#!/usr/bin/env python3
"""Generates tests/golden/ref_callers.npz: what the reference's CALLERS compute between forward
and backward pass, on the REAL reference net (oracle/_ref/librecur_ref.so), for the three losses the
oracle restates without a compiled counterpart -- their files need the generated path.h / GStreamer:

  * multi_softmax_error + text_train   charmodel-multi-predict.c:19-58, 234-281
  * train_channel's grouped softmax      gstclassify.c:2070-2127
  * train_net's sigmoid-slope error      gstrnnca.c:693-716

Each is a few lines of control flow around primitives that ARE the reference's own object code
(rnn_opinion, rnn_bptt_advance / _calc_deltas / rnn_apply_learning from recur-nn.c; softmax_best_guess,
rand64, fast_sigmoid from its headers through oracle/ref_shim.c).  This script walks that control
flow in Python, calling those primitives on the reference's own structs, and records inputs and
results; tests/test_oracle_callers_golden.py holds orc_multi_softmax_error / orc_multitext_train /
orc_grouped_softmax_error / orc_sigmoid_mse_error to them.  Run where /root/reference exists:

    make -C oracle && python tests/golden/make_golden_callers.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import callers_cases as cc  # noqa: E402
import recur_ctypes as rc  # noqa: E402
import scenarios as sc  # noqa: E402

U64_MAX_AS_FLOAT = np.float32(2.0 ** 64)  # (float)UINT64_MAX


def multi_softmax_error(ref, a, c, nxt, target_class, alen, leakage):
    """charmodel-multi-predict.c:19-58 on stream 0 of the reference set `a`: returns (err, ranges)."""
    net = a.nets[0]
    n = net.contents
    answer = a.one_hot_opinion(0, c, n.presynaptic_noise)
    err_p = n.bptt.contents.o_error
    error = rc.view(err_p, a.O)
    n_classes = a.output_size // alen
    # `u64 threshold = leakage * UINT64_MAX;` -- a float product converted to u64
    threshold = int(np.float32(leakage) * U64_MAX_AS_FLOAT)
    error[:a.output_size] = 0
    ranges, err = [], 0.0
    for i in range(n_classes):
        off = i * alen
        if i == target_class or ref.ref_rand64(C.byref(n.rng)) < threshold:
            ref.ref_softmax_best_guess(C.cast(C.addressof(err_p.contents) + 4 * off, C.POINTER(C.c_float)),
                                       C.cast(C.addressof(answer.contents) + 4 * off, C.POINTER(C.c_float)), alen)
            error[off + nxt] += 1.0
            if i == target_class:
                err = float(error[off + nxt])
            start, end = off & ~3, (off + alen + 3) & ~3
            if ranges and ranges[-1][0] + ranges[-1][1] >= start:
                ranges[-1][1] = end - ranges[-1][0]
                continue
            ranges.append([start, end - start])
    return err, ranges


def text_train(ref, a, text, target_class, alen, leakage, method, batch, trace):
    """charmodel-multi-predict.c:234-281 (the loop of text_train) on the reference net."""
    net = a.net
    bptt = net.contents.bptt.contents
    countdown = batch - net.contents.generation % batch
    err_sum = ent_sum = np.float32(0)
    for i in range(len(text) - 1):
        ref.rnn_bptt_advance(net)
        e, ranges = multi_softmax_error(ref, a, int(text[i]), int(text[i + 1]), target_class, alen, leakage)
        r = (rc.ErrorRange * (len(ranges) + 1))(*[tuple(x) for x in ranges], (-1, 0))
        if countdown == 0:
            ref.rnn_apply_learning(net, method, bptt.momentum)
            countdown = batch
            ref.rnn_bptt_calc_deltas(net, 0, r)
        else:
            ref.rnn_bptt_calc_deltas(net, 1, r)
        countdown -= 1
        err_sum = np.float32(err_sum + np.float32(e))
        ent_sum = np.float32(ent_sum + np.float32(-100.0 if 1.0 - e < 1e-30 else np.log2(np.float32(1.0 - e))))
        trace.append((e, [x for rr in ranges for x in rr]))
    return float(err_sum), float(ent_sum)


def main():
    ref = rc.load_ref()
    out = {}

    # ---- the multi-head trainer ------------------------------------------------------------
    for name, c in cc.MULTI_CASES.items():
        a = sc.ApiSet(ref, softmax_best_guess=ref.ref_softmax_best_guess, **cc.multi_kwargs(c))
        if c["method"] == rc.ADAGRAD:  # its accumulators are the momentum arrays (recur-nn.c:617-621): the ballast
            ref.rnn_set_momentum_values(a.net, c["ballast"])
        a.net.contents.bptt.contents.momentum = 0.9
        text = cc.multi_text(c)
        trace = []
        for cls, lo, hi in cc.multi_passes(c, len(text)):
            es, hs = text_train(ref, a, text[lo:hi], cls, c["A"], c["leakage"], c["method"], c["batch"], trace)
            out.setdefault(name + ".sums", []).append([es, hs])
        out[name + ".sums"] = np.array(out[name + ".sums"], np.float32)
        out[name + ".err"] = np.array([t[0] for t in trace], np.float32)
        width = 2 * (c["classes"] + 1)
        out[name + ".ranges"] = np.array([t[1] + [-1] * (width - len(t[1])) for t in trace], np.int32)
        for k, v in a.snapshot().items():
            out["%s.%s" % (name, k)] = v
        # rnn_char_multi_cross_entropy (charmodel-multi-predict.c:383-408) on the trained net: per head, bits per
        # symbol of the first 60 symbols, the first 5 only fed
        probe, skip = text[:60], 5
        ent = np.zeros(c["classes"], np.float64)
        p = np.zeros(c["A"], np.float32)
        for i in range(len(probe) - 1):
            answer = a.one_hot_opinion(0, int(probe[i]), 0.0)
            if i < skip:
                continue
            for h in range(c["classes"]):
                ref.ref_softmax(rc.fptr(p), C.cast(C.addressof(answer.contents) + 4 * h * c["A"], C.POINTER(C.c_float)),
                                c["A"])
                e = np.float32(p[int(probe[i + 1])])
                ent[h] -= np.float64(np.float32(-100.0) if e < 1e-30 else np.log2(e))
        out[name + ".xent"] = ent / (len(probe) - skip - 1)
        a.close()

    # ---- gstclassify's channel loss ----------------------------------------------------------
    for name, c in cc.GROUP_CASES.items():
        answers, targets, weight = cc.group_inputs(c)
        O = int(sum(c["sizes"]))
        errors = np.zeros_like(answers)
        stats = []
        for s in range(len(answers)):
            err = errors[s]
            wins, wrong, trained = 0, np.float32(0), 0
            off = 0
            for g, n_cls in enumerate(c["sizes"]):
                t = int(targets[s, g])
                if t < 0 or t >= n_cls:
                    err[off:off + n_cls] = 0
                else:
                    w = ref.ref_softmax_best_guess(rc.fptr(err[off:]), rc.fptr(answers[s, off:]), n_cls)
                    wins += int(w == t)
                    err[off + t] += 1.0
                    wrong = np.float32(wrong + err[off + t])
                    trained += 1
                off += n_cls
            if trained and weight is not None:
                err[:O] *= weight
            stats.append([trained, wins, float(wrong)])
        out[name + ".answers"], out[name + ".targets"] = answers, targets
        out[name + ".errors"] = errors
        out[name + ".stats"] = np.array(stats, np.float64)
        if weight is not None:
            out[name + ".weight"] = weight

    # ---- rnnca's loss --------------------------------------------------------------------------
    answers, targets = cc.sigmoid_inputs()
    sig = np.array([[ref.ref_fast_sigmoid(float(x)) for x in row] for row in answers], np.float32)
    slope = sig * (np.float32(1.0) - sig)
    out["sigmoid.answers"], out["sigmoid.targets"] = answers, targets
    out["sigmoid.activated"] = sig
    out["sigmoid.errors"] = (slope * (targets - sig)).astype(np.float32)

    path = os.path.join(HERE, "ref_callers.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%d arrays, %.1f KB" % (len(out), os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
