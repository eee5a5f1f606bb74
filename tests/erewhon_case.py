"""The erewhon.txt training run shared by golden/make_golden.py (real reference),
the oracle tests and the GPU tests: text-predict's multi-tap epoch loop
(charmodel-predict.c:288-311) with 99 hidden units, 4 streams, BPTT depth 10,
learn rate 1e-3, weighted momentum 0.95, on the default alphabet; training
entropy per 250-generation window and the validation cross-entropy of
get_cross_entropy (charmodel-predict.c:62-80) over the last 4000 symbols, through
a weight-borrowing forward-only clone (text-predict.c:538-541)."""
import ctypes as C

import numpy as np

import recur_ctypes as rc
import scenarios as sc

GENERATIONS, WINDOW, VALIDATE = 1500, 250, 4000
KW = dict(input_size=42, hidden_size=99, output_size=42, S=4, D=10, learn_rate=1e-3, seed=1)
_text = None


def encoded_text():
    global _text
    if _text is None:
        _text = rc.encode_erewhon(rc.load_amd())
    return _text


def capped_log2(x):
    return -100.0 if x < 1e-30 else float(np.log2(np.float32(x)))


def cross_entropy(lib, net, softmax, text, skip=5):
    """get_cross_entropy (charmodel-predict.c:62-80) on a forward-only clone"""
    clone = lib.rnn_clone(net, net.contents.flags & ~(rc.FLAG_OWN_BPTT | rc.FLAG_OWN_WEIGHTS), rc.SUBSEED, None)
    n = clone.contents
    probs = np.zeros(n.o_size, np.float32)
    total = 0.0
    for i in range(len(text) - 1):
        real = rc.view(n.real_inputs, n.input_size)
        real[:] = 0
        real[int(text[i])] = 1.0
        ans = lib.rnn_opinion(clone, None, 0.0)
        if i >= skip:
            softmax(rc.fptr(probs), ans, n.output_size)
            total += capped_log2(probs[int(text[i + 1])])
    lib.rnn_delete_net(clone)
    return total / -(len(text) - skip - 1)


def run(lib, softmax_best_guess, softmax, batched):
    text = encoded_text()
    train, valid = text[:-VALIDATE], np.ascontiguousarray(text[-VALIDATE:])
    train = np.ascontiguousarray(train)
    if batched:
        a = sc.AmdBatchedSet(lib, **KW)
        a.load_text(train)
    else:
        a = sc.ApiSet(lib, softmax_best_guess=softmax_best_guess, **KW)
    windows = []
    ent = 0.0
    for i in range(GENERATIONS):
        if batched:
            lib.rnn_amd_set_char_step(a.handle, i, rc.WEIGHTED, 0.95)
        else:
            for e, _c in a.char_step(train, i, rc.WEIGHTED, 0.95):
                ent += capped_log2(1.0 - e)
        if (i + 1) % WINDOW == 0:
            if batched:
                st = a.stats(clear=True)
                ent = st.entropy
            windows.append(-ent / (WINDOW * a.S))
            ent = 0.0
    ventropy = cross_entropy(lib, a.net, softmax, valid)
    a.close()
    return {"t_entropy": np.array(windows), "v_entropy": np.array([ventropy])}


def run_oracle():
    text = encoded_text()
    train, valid = np.ascontiguousarray(text[:-VALIDATE]), np.ascontiguousarray(text[-VALIDATE:])
    o = sc.OracleSet(**KW)
    z = o.z.contents
    windows = []
    for i in range(GENERATIONS):
        o.char_step(train, i, rc.WEIGHTED, 0.95)
        if (i + 1) % WINDOW == 0:
            windows.append(-z.stat_entropy / (WINDOW * o.S))
            z.stat_entropy = 0.0
    # validation on a fresh single-stream set sharing the trained weights
    v = sc.OracleSet(input_size=42, hidden_size=99, output_size=42, S=1, D=1, learn_rate=1e-3, seed=1)
    for k in ("ih_w", "ho_w"):
        v.arrays()[k][:] = o.arrays()[k]
    ventropy = v.orc.orc_cross_entropy(v.z, 0, rc.u8ptr(valid), len(valid), 5)
    o.close()
    v.close()
    return {"t_entropy": np.array(windows), "v_entropy": np.array([ventropy])}
