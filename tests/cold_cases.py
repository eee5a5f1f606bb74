"""The cold host-side functions of recur-nn.h (weight noise, perforation, diagonal
zapping, gain scaling, the FAN_IN / RUNS / ZERO initialisers, optimiser ballast,
rnn_log_net) as one replayable script over the recur-nn.h ABI.

``run(lib)`` drives any library that speaks that ABI and returns name -> array.
tests/golden/make_golden_cold.py runs it on the REAL reference
(oracle/_ref/librecur_ref.so) and stores the result as tests/golden/ref_cold.npz;
tests/test_abi.py replays it on librecur_amd.so (host-only calls: no GPU needed)
and demands bit-exact equality, generator states included.

Reference: recur-nn.c:857-904, 1027-1145; recur-nn-init.c:359-380, 382-861.
"""
import ctypes as C
import os
import tempfile

import numpy as np

import recur_ctypes as rc

FLAGS = rc.FLAG_STANDARD | rc.FLAG_ADAPTIVE_MIN_ERROR


def _rng(n):
    return np.array([n.rng.a, n.rng.b, n.rng.c, n.rng.d], dtype=np.uint64)


def _weights(out, name, net):
    n = net.contents
    out[name + ".ih_w"] = rc.view(n.ih_weights, n.i_size, n.h_size).copy()
    out[name + ".ho_w"] = rc.view(n.ho_weights, n.h_size, n.o_size).copy()
    out[name + ".rng"] = _rng(n)


def _new(lib, seed, inputs=10, hidden=31, outputs=5, flags=FLAGS, depth=4):
    net = lib.rnn_new(inputs, hidden, outputs, flags, seed, None, depth, 0.01, 0.9, 0.0, rc.RELU)
    lib.rnn_randomise_weights_auto(net)
    return net


def run(lib):
    out = {}
    # rnn_weight_noise (recur-nn.c:857-883), with and without a bottom layer
    net = _new(lib, 21)
    lib.rnn_weight_noise(net, 0.01)
    _weights(out, "noise", net)
    lib.rnn_weight_noise(net, 0.5)  # continues the same generator
    _weights(out, "noise2", net)
    lib.rnn_delete_net(net)
    net = lib.rnn_new_with_bottom_layer(12, 6, 20, 5, FLAGS, 22, None, 4, 0.01, 0.9, 0.0, rc.RELU, 0)
    lib.rnn_randomise_weights_auto(net)
    lib.rnn_weight_noise(net, 0.02)
    _weights(out, "noise_bottom", net)
    bl = net.contents.bottom_layer.contents
    out["noise_bottom.b_w"] = rc.view(bl.weights, bl.i_size, bl.o_size).copy()
    lib.rnn_delete_net(net)

    # rnn_perforate_weights (recur-nn-init.c:739-742; recur-nn-helpers.h:84-102): the
    # general path and the dropout == 0.5 bit path
    for name, p in (("perforate30", 0.3), ("perforate50", 0.5)):
        net = _new(lib, 23)
        lib.rnn_perforate_weights(net, p)
        _weights(out, name, net)
        lib.rnn_delete_net(net)

    # rnn_zap_non_diagonals / rnn_clear_diagonal_only_section (recur-nn.c:1082-1145).
    # zap_2_20_3 takes the "friend parameter is stupid" branch, which leaves friend_start
    # as it was (rows 0 and 1 keep one weight each).  Not a case: friends > stop - start
    # with start - friends < -(stop - start), where the reference's memset length goes
    # negative and it crashes (e.g. 10, 20, 30); the product zeroes safely there.
    for name, args in (("zap_5_20_3", (5, 20, 3)), ("zap_1_32_0", (1, 32, 0)), ("zap_8_40_2", (8, 40, 2)),
                       ("zap_2_20_3", (2, 20, 3)), ("zap_40_50_0", (40, 50, 0))):
        net = _new(lib, 24)
        lib.rnn_zap_non_diagonals(net, *args)
        _weights(out, name, net)
        lib.rnn_delete_net(net)
    for name, args in (("cleardiag_10_2", (10, 2)), ("cleardiag_6_9", (6, 9)), ("cleardiag_0_0", (0, 0))):
        net = _new(lib, 25)
        lib.rnn_clear_diagonal_only_section(net, *args)
        _weights(out, name, net)
        lib.rnn_delete_net(net)

    # rnn_scale_initial_weights (recur-nn.c:1027-1076)
    net = _new(lib, 26, hidden=23)
    lib.rnn_scale_initial_weights(net, 0.8)
    _weights(out, "scale_gain", net)
    lib.rnn_delete_net(net)

    # the other initialisers (recur-nn-init.c:382-683)
    def init(name, seed, tweak, hidden=31):
        net = lib.rnn_new(10, hidden, 5, FLAGS, seed, None, 4, 0.01, 0.9, 0.0, rc.RELU)
        p = rc.InitParams()
        lib.rnn_init_default_weight_parameters(net, C.byref(p))
        tweak(p)
        lib.rnn_randomise_weights_clever(net, C.byref(p))
        _weights(out, name, net)
        lib.rnn_delete_net(net)

    def fan_in(p):
        p.method = rc.INIT_FAN_IN

    def fan_in_sharp(p):
        p.method = rc.INIT_FAN_IN
        p.fan_in_sum, p.fan_in_step, p.fan_in_min = 2.0, 0.5, 0.05

    def zero(p):
        p.method = rc.INIT_ZERO

    def runs(p):
        p.method = rc.INIT_RUNS

    def runs_sub(p):
        p.method = rc.INIT_RUNS
        p.submethod = rc.INIT_FLAT
        p.bias_uses_submethod = 1
        p.inputs_use_submethod = 1

    def runs_bias(p):
        p.method = rc.INIT_RUNS
        p.submethod = rc.INIT_FAN_IN
        p.bias_uses_submethod = 1

    def runs_open(p):
        p.method = rc.INIT_RUNS
        p.run_loop = 0
        p.run_crossing_paths = 1
        p.run_inputs_miss = 1
        p.run_input_at_start = 1
        p.run_n = 5
        p.run_len_mean = 9.0
        p.run_len_stddev = 2.0

    init("init_fan_in", 27, fan_in)
    init("init_fan_in_sharp", 28, fan_in_sharp, hidden=64)
    init("init_zero", 29, zero)
    init("init_runs", 30, runs)
    init("init_runs_sub", 31, runs_sub)
    init("init_runs_bias", 32, runs_bias)
    init("init_runs_open", 33, runs_open, hidden=64)

    # optimiser ballast (recur-nn-init.c:359-380)
    net = _new(lib, 34, flags=FLAGS | rc.FLAG_AUX_ARRAYS)
    lib.rnn_set_momentum_values(net, 0.25)
    lib.rnn_set_aux_values(net, 1e-4)
    n = net.contents
    b = n.bptt.contents
    out["ballast.ih_m"] = rc.view(b.ih_momentum, n.ih_size).copy()
    out["ballast.ho_m"] = rc.view(b.ho_momentum, n.ho_size).copy()
    out["ballast.ih_aux"] = rc.view(b.ih_aux, n.ih_size).copy()
    out["ballast.ho_aux"] = rc.view(b.ho_aux, n.ho_size).copy()
    lib.rnn_delete_net(net)

    # rnn_log_net and rnn_set_log_file's first line (recur-nn.c:887-904; recur-nn-init.c:268-283)
    net = _new(lib, 35)
    n = net.contents
    b = n.bptt.contents
    rs = np.random.default_rng(3)
    rc.view(b.o_error, n.o_size)[:] = rs.standard_normal(n.o_size).astype(np.float32)
    rc.view(b.h_error, n.i_size)[:] = rs.standard_normal(n.i_size).astype(np.float32) * 3
    n.generation = 77
    fd, path = tempfile.mkstemp(suffix=".log")
    os.close(fd)
    lib.rnn_set_log_file(net, path.encode(), 0)
    lib.rnn_log_net(net)
    lib.rnn_set_log_file(net, None, 0)
    out["log_net.lines"] = np.frombuffer(open(path, "rb").read(), dtype=np.uint8).copy()
    os.unlink(path)
    lib.rnn_delete_net(net)
    return out
