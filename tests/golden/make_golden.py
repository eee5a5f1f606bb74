#!/usr/bin/env python3
"""Generates tests/golden/ref_vectors.npz by running the REAL reference
(oracle/_ref/librecur_ref.so, compiled from /root/reference by oracle/Makefile)
through ctypes.  Run it in the container that has /root/reference:

    make -C oracle && python tests/golden/make_golden.py

The file it writes is pure data (inputs and the reference's outputs); nothing of
the reference's source travels.  tests/test_oracle_golden.py pins the oracle to
it and tests/test_gpu_parity.py pins librecur_amd.so to it.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import golden_cases as gc  # noqa: E402
import recur_ctypes as rc  # noqa: E402
import scenarios as sc  # noqa: E402


def main():
    ref = rc.load_ref()
    out = {}

    # G1: PRNG and small maths --------------------------------------------
    for seed in (1, 2, 11):
        ctx = rc.RandCtx()
        ref.ref_init_rand64(C.byref(ctx), seed)
        out["rand64_seed%d" % seed] = np.array([ref.ref_rand64(C.byref(ctx)) for _ in range(64)],
                                               dtype=np.uint64)
    ctx = rc.RandCtx()
    ref.ref_init_rand64(C.byref(ctx), 1)
    out["gaussian_seed1"] = np.array([ref.ref_cheap_gaussian_noise(C.byref(ctx)) for _ in range(32)],
                                     dtype=np.float32)
    out["rand_double_seed1"] = np.array([ref.ref_rand_double(C.byref(ctx)) for _ in range(16)])
    out["rand_small_int_seed1"] = np.array([ref.ref_rand_small_int(C.byref(ctx), 1000)
                                            for _ in range(16)], dtype=np.int32)
    xs = np.concatenate([np.linspace(-70, 60, 261), [0.0, 0.2, -0.2, 0.2000001, 1e-9]]).astype(np.float32)
    out["expf_x"] = xs
    out["expf_y"] = np.array([ref.ref_fast_expf(float(x)) for x in xs], dtype=np.float32)
    rng = np.random.default_rng(5)
    sm_in = (rng.standard_normal((6, 44)) * np.array([1, 5, 20, 40, 0.1, 80])[:, None]).astype(np.float32)
    sm_in[5, 3] = 95.0
    sm_out = np.zeros_like(sm_in)
    sm_err = np.zeros_like(sm_in)
    best = np.zeros(6, np.int32)
    for i in range(6):
        ref.ref_softmax(rc.fptr(sm_out[i]), rc.fptr(sm_in[i]), 42)
        best[i] = ref.ref_softmax_best_guess(rc.fptr(sm_err[i]), rc.fptr(sm_in[i]), 42)
    out.update(softmax_in=sm_in, softmax_out=sm_out, softmax_err=sm_err, softmax_best=best)
    sums = np.array([0.5, 1.9, 2.0, 2.1, 5.0, 40.0, 1e4], np.float32)
    out["soft_clip_in"] = sums
    out["soft_clip_out"] = np.array([ref.ref_soft_clip(float(s), 2.0) for s in sums], dtype=np.float32)
    out["soft_start"] = np.array([ref.rnn_calculate_momentum_soft_start(g, 0.95, x)
                                  for g in (0, 10, 1000, 1e6) for x in (0, 1, 2000)], dtype=np.float32)

    # strings around the character models: the hash in net file names (recur-common.h:207-216),
    # UTF-8 in both directions (utf8.h), the confabulation softmax (badmaths.h:143-156)
    samples = [b"", b"a", b"alphabet 8%20etaonihsrdlucmwfygpb,v.k-;x\"qj'?:z)(_!*&\ncollapse_chars "
               b"10872}{659/34][@\nutf8 0\ncollapse_space 1\ncase_insensitive 1\n",
               bytes(range(1, 256)), b"The quick brown fox" * 7]
    out["hash32_inputs"] = np.frombuffer(b"\0".join(samples), dtype=np.uint8).copy()
    out["hash32_input_lens"] = np.array([len(x) for x in samples], np.int32)
    out["hash32_values"] = np.array([ref.ref_hash32(x) for x in samples], dtype=np.uint32)
    codes = [0x24, 0x7f, 0x80, 0xe9, 0x7ff, 0x800, 0x20ac, 0xffff, 0x10000, 0x1f600, 0x1fffff, 0x200000]
    enc = []
    for code in codes:
        buf = C.create_string_buffer(8)
        n = ref.ref_write_utf8_char(code, buf)
        enc.append([n] + list(buf.raw[:4]))
    out["utf8_codes"] = np.array(codes, np.uint32)
    out["utf8_encoded"] = np.array(enc, np.int32)
    seqs = [b"A", b"\xc3\xa9", b"\xe2\x82\xac", b"\xf0\x9f\x98\x80", b"\x80", b"\xc0\xaf", b"\xe0\x80\xaf",
            b"\xf8\x88\x80\x80", b"\xc3", b"\xe2\x82", b"\xf0\x80\x80\x80", b"\xdf\xbf", b"\xef\xbf\xbf"]
    dec = []
    for q in seqs:
        used = C.c_int(0)
        dec.append([ref.ref_read_utf8_char(q + b"\0\0\0\0", C.byref(used)), used.value])
    out["utf8_sequences"] = np.array([list(q.ljust(4, b"\0")) for q in seqs], np.uint8)
    out["utf8_decoded"] = np.array(dec, np.int32)
    bs_in = (np.random.default_rng(8).standard_normal(42) * 3).astype(np.float32)
    out["biased_softmax_in"] = bs_in
    for bias in (0.0, 1.0, 7.5):
        o = np.zeros(42, np.float32)
        ref.ref_biased_softmax(rc.fptr(o), rc.fptr(bs_in), 42, bias)
        out["biased_softmax_%g" % bias] = o

    # G2: weight initialisation ------------------------------------------------
    for name, (hidden, shape, perf, seed) in {
        "init_semicircle_h99": (99, rc.DIST_SEMICIRCLE, 0.0, 1),
        "init_uniform_perf_h99": (99, rc.DIST_UNIFORM, 0.7, 1),
        "init_gaussian_h1024": (1024, rc.DIST_GAUSSIAN, 0.0, 2),
        "init_lognormal_h64": (64, rc.DIST_LOG_NORMAL, 0.2, 3),
    }.items():
        net = ref.rnn_new(42, hidden, 42, rc.FLAG_STANDARD, seed, None, 10, 1e-3, 0.9, 0.0, rc.RELU)
        p = rc.InitParams()
        ref.rnn_init_default_weight_parameters(net, C.byref(p))
        p.method, p.flat_shape, p.flat_perforation = rc.INIT_FLAT, shape, perf
        ref.rnn_randomise_weights_clever(net, C.byref(p))
        n = net.contents
        ih = rc.view(n.ih_weights, n.ih_size)
        ho = rc.view(n.ho_weights, n.ho_size)
        out[name] = np.array([ih.astype(np.float64).sum(), np.abs(ih).astype(np.float64).sum(),
                              ho.astype(np.float64).sum(), np.abs(ho).astype(np.float64).sum()])
        out[name + "_ih_head"] = ih[n.h_size:n.h_size + 256].copy()
        out[name + "_ho_head"] = ho[:256].copy()
        out[name + "_rng"] = np.array([n.rng.a, n.rng.b, n.rng.c, n.rng.d], dtype=np.uint64)
        ref.rnn_delete_net(net)

    # ... and with a bottom layer, whose weights are drawn last (recur-nn-init.c:566-572)
    for name, method in (("init_bottom_flat", rc.INIT_FLAT), ("init_bottom_fan_in", rc.INIT_FAN_IN)):
        net = ref.rnn_new_with_bottom_layer(42, 16, 39, 42, rc.FLAG_STANDARD, 5, None, 10, 1e-3, 0.9,
                                            0.0, rc.RELU, 0)
        p = rc.InitParams()
        ref.rnn_init_default_weight_parameters(net, C.byref(p))
        p.method = method
        ref.rnn_randomise_weights_clever(net, C.byref(p))
        n = net.contents
        bl = n.bottom_layer.contents
        out[name + "_b_w"] = rc.view(bl.weights, bl.i_size, bl.o_size).copy()
        out[name + "_ih_sum"] = np.array([rc.view(n.ih_weights, n.ih_size).astype(np.float64).sum()])
        out[name + "_rng"] = np.array([n.rng.a, n.rng.b, n.rng.c, n.rng.d], dtype=np.uint64)
        ref.rnn_delete_net(net)

    # G3/G4: training scenarios -----------------------------------------------------
    text = sc.synthetic_text(gc.TEXT_LEN)
    assert np.array_equal(text, sc.synthetic_text_oracle(gc.TEXT_LEN))
    for name, c in gc.TRAIN_CASES.items():
        a = sc.ApiSet(ref, softmax_best_guess=ref.ref_softmax_best_guess, **gc.case_kwargs(c))
        gc.prepare(a, c, lambda x: ref.rnn_set_momentum_values(a.net, x),
                   lambda x: ref.rnn_set_aux_values(a.net, x))
        depth_trace, scale_trace = [], []

        def trace(i):
            scale_trace.append([a.nets[j].contents.bptt.contents.ih_scale for j in range(a.S)])
        gc.run_scenario(a, c, text, trace)
        snap = a.snapshot()
        b0 = a.net.contents.bptt.contents
        for k, v in snap.items():
            out["%s.%s" % (name, k)] = v
        if c.get("aux"):
            out[name + ".ih_aux"] = rc.view(b0.ih_aux, a.I, a.H).copy()
            out[name + ".ho_aux"] = rc.view(b0.ho_aux, a.H, a.O).copy()
        out[name + ".h_error"] = np.stack([rc.view(a.nets[j].contents.bptt.contents.h_error, a.I)
                                           for j in range(a.S)])
        out[name + ".i_error"] = np.stack([rc.view(a.nets[j].contents.bptt.contents.i_error, a.I)
                                           for j in range(a.S)])
        out[name + ".ih_scale_trace"] = np.array(scale_trace, np.float32)
        a.close()

    # G5: conditioning, one bit at a time (recur-nn.c:782-855) ---------------------------
    for bit, flag in ((0, rc.COND_USE_SCALE), (2, rc.COND_USE_ZERO), (3, rc.COND_USE_LAWN_MOWER),
                      (4, rc.COND_USE_TALL_POPPY), (6, rc.COND_USE_RAND)):
        net = ref.rnn_new(10, 31, 5, rc.FLAG_OWN_BPTT | rc.FLAG_OWN_WEIGHTS | flag, 21, None, 4, 0.01,
                          0.9, 0.0, rc.RELU)
        ref.rnn_randomise_weights_auto(net)
        n = net.contents
        w = rc.view(n.ih_weights, n.ih_size)
        w[7] = 14.5
        w[11] = -12.0
        w[13] = 1e-36
        rc.view(n.bptt.contents.ih_momentum, n.ih_size)[5] = 1e-37
        n.generation = 8 + bit
        ref.rnn_condition_net(net)
        out["cond%d.ih_w" % bit] = w.copy()
        out["cond%d.ho_w" % bit] = rc.view(n.ho_weights, n.ho_size).copy()
        out["cond%d.ih_m" % bit] = rc.view(n.bptt.contents.ih_momentum, n.ih_size).copy()
        ref.rnn_delete_net(net)

    # the single-net fused path, rnn_bptt_calculate (recur-nn.c:999-1019) ----------------
    for name, batch in (("fused_b1", 1), ("fused_b4", 4)):
        a = sc.ApiSet(ref, softmax_best_guess=ref.ref_softmax_best_guess, input_size=42, hidden_size=39,
                      output_size=42, S=1, D=8, learn_rate=1e-2, seed=13)
        for i in range(24):
            ref.rnn_bptt_advance(a.net)
            a.net_error_bptt(0, int(text[i]), int(text[i + 1]))
            a.net.contents.bptt.contents.momentum = 0.9
            ref.rnn_bptt_calculate(a.net, batch)
        for k, v in a.snapshot().items():
            out["%s.%s" % (name, k)] = v
        a.close()

    # sparse top layer with error ranges (recur-nn.c:156-196, 275-301) ------------------------
    a = sc.ApiSet(ref, softmax_best_guess=ref.ref_softmax_best_guess, input_size=20, hidden_size=39,
                  output_size=24, S=3, D=6, learn_rate=1e-2, seed=14)
    ranges = (rc.ErrorRange * 3)((4, 8), (16, 4), (-1, 0))
    rs = np.random.default_rng(9)
    for i in range(10):
        for j in range(3):
            ref.rnn_bptt_advance(a.nets[j])
            x = rs.standard_normal(20).astype(np.float32) * (rs.random(20) < 0.5)
            ref.rnn_opinion(a.nets[j], rc.fptr(x.astype(np.float32)), 0.0)
            e = rc.view(a.nets[j].contents.bptt.contents.o_error, a.O)
            e[:] = 0
            e[4:12] = rs.standard_normal(8) * 0.1
            e[16:20] = rs.standard_normal(4) * 0.1
            ref.rnn_bptt_calc_deltas(a.nets[j], 1 if j else 0, ranges)
        ref.rnn_apply_learning(a.net, rc.WEIGHTED, 0.9)
    for k, v in a.snapshot().items():
        out["sparse.%s" % k] = v
    out["sparse.h_error"] = np.stack([rc.view(a.nets[j].contents.bptt.contents.h_error, a.I)
                                      for j in range(3)])
    a.close()

    # text-predict on erewhon.txt (BASELINE.json configs[0] shape: 99 hidden), multi-tap 4:
    # training entropy per window and the validation cross-entropy (G7) ------------------
    import erewhon_case as ec
    res = ec.run(ref, ref.ref_softmax_best_guess, ref.ref_softmax, batched=False)
    for k, v in res.items():
        out["erewhon." + k] = v

    path = os.path.join(HERE, "ref_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%d arrays, %.1f KB" % (len(out), os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
