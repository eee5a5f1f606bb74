"""Small drivers of librecur_amd.so's C API that benchmarks, tools and tests share.

``ApiSet`` drives a library that speaks the recur-nn.h ABI (librecur_amd.so through its per-net drop-in calls; the
tests also point it at the compiled reference) exactly the way the reference's own caller does (rnn_char_epoch,
charmodel-predict.c:288-311).  ``AmdBatchedSet`` drives the additive batched entry points (include/recur_amd.h
part 2): the whole set per call, text on the device.  ``synthetic_text_np`` is the seeded symbol stream of
SURVEY.md section 8(d)'s data-free workload, restated in numpy (Jenkins PRNG, recur-rng.h) so that input data
never comes out of a checker's library.
"""
import ctypes as C

import numpy as np

from . import api as rc



def synthetic_text_np(n=6000, alphabet=42, seed=7):
    """The seeded symbol stream of scenarios.synthetic_text, restated in numpy so
    that fixtures do not depend on any C library (Jenkins PRNG, recur-rng.h)."""
    M = (1 << 64) - 1

    def rot(x, k):
        return ((x << k) | (x >> (64 - k))) & M

    a, b, c, d = 0xF1EA5EED, seed, seed, seed

    def step():
        nonlocal a, b, c, d
        e = (a - rot(b, 7)) & M
        a = b ^ rot(c, 13)
        b = (c + rot(d, 37)) & M
        c = (d + e) & M
        d = (e + a) & M
        return d

    for _ in range(20):
        step()
    out = np.empty(n, np.uint8)
    for i in range(n):
        bits = (step() & 0x000FFFFFFFFFFFFF) | 0x3FF0000000000000
        x = np.frombuffer(np.uint64(bits).tobytes(), dtype=np.float64)[0] - 1.0
        out[i] = int(x * alphabet)
    return out



class ApiSet:
    """A training set held by an rnn_* library (reference or product)."""

    def __init__(self, lib, input_size, hidden_size, output_size, S, D, activation=rc.RELU,
                 flags=rc.FLAG_STANDARD | rc.FLAG_ADAPTIVE_MIN_ERROR, learn_rate=1e-3, seed=1,
                 momentum=0.95, variance=None, shape=rc.DIST_SEMICIRCLE, perforation=0.0,
                 softmax_best_guess=None, noise=0.0, bottom_inputs=0, bottom_rate_scale=1.0,
                 shard=None):
        self.lib = lib
        self.S, self.D = S, D
        # shard = (global_first, global_count): this set holds S of the streams of a larger
        # logical set (one shard per GPU / process)
        self.global_first, self.global_count = shard if shard else (0, S)
        # with bottom_inputs the net sits on a bottom layer: `input_size` is that
        # layer's output size and bottom_inputs its input size
        # (rnn_new_with_bottom_layer, recur-nn-init.c:194-219)
        self.bottom_inputs = bottom_inputs
        if bottom_inputs:
            net = lib.rnn_new_with_bottom_layer(bottom_inputs, input_size, hidden_size, output_size,
                                                flags, seed, None, D, learn_rate, momentum, noise,
                                                activation, 0)
            net.contents.bottom_layer.contents.learn_rate_scale = bottom_rate_scale
        else:
            net = lib.rnn_new(input_size, hidden_size, output_size, flags, seed, None, D,
                              learn_rate, momentum, noise, activation)
        self.net = net
        n = net.contents
        self.I, self.H, self.O = n.i_size, n.h_size, n.o_size
        self.input_size, self.hidden_size, self.output_size = input_size, hidden_size, output_size
        p = rc.InitParams()
        lib.rnn_init_default_weight_parameters(net, C.byref(p))
        p.method = rc.INIT_FLAT
        if variance is not None:
            p.flat_variance = variance
        p.flat_shape = shape
        p.flat_perforation = perforation
        lib.rnn_randomise_weights_clever(net, C.byref(p))
        if shard and shard != (0, S):
            self.nets = lib.rnn_amd_new_training_set_shard(net, S, shard[0], shard[1])
        else:
            self.nets = lib.rnn_new_training_set(net, S)
        # the caller-side loss (softmax_best_guess is a static inline of the
        # reference's badmaths.h, i.e. caller code, not library code)
        self._sbg = softmax_best_guess

    def close(self):
        self.lib.rnn_delete_training_set(self.nets, self.S, 0)

    # -- per-stream steps (the reference call sequence) --
    def one_hot_opinion(self, j, hot, noise=0.0):
        n = self.nets[j].contents
        if self.bottom_inputs:
            # the helper's bottom-layer branch indexes from the bias slot
            # (charmodel-helpers.h:20-23, 30-31)
            real = rc.view(n.bottom_layer.contents.inputs, self.bottom_inputs)
        else:
            real = rc.view(n.real_inputs, self.input_size)
        real[:] = 0
        real[hot] = 1.0
        return self.lib.rnn_opinion(self.nets[j], None, noise)

    def net_error_bptt(self, j, c, nxt):
        n = self.nets[j].contents
        answer = self.one_hot_opinion(j, c, n.presynaptic_noise)
        err = n.bptt.contents.o_error
        winner = self._sbg(err, answer, self.output_size)
        e = rc.view(err, self.O)
        e[nxt] += 1.0
        return float(e[nxt]), int(winner == nxt)

    def char_step_deltas(self, text, i):
        L = len(text)
        spacing = (L - 1) // self.global_count
        stats = []
        for j in range(self.S):
            off = i + (self.global_first + j) * spacing
            if off >= L - 1:
                off -= L - 1
            self.lib.rnn_bptt_advance(self.nets[j])
            stats.append(self.net_error_bptt(j, int(text[off]), int(text[off + 1])))
            self.lib.rnn_bptt_calc_deltas(self.nets[j], 1 if j else 0, None)
        return stats

    def char_step(self, text, i, method=rc.WEIGHTED, momentum=0.95):
        stats = self.char_step_deltas(text, i)
        self.lib.rnn_apply_learning(self.net, method, momentum)
        return stats

    def sync(self):
        if hasattr(self.lib, "rnn_amd_sync_host"):
            self.lib.rnn_amd_sync_host(self.net, rc.RNN_AMD_EVERYTHING)

    def snapshot(self):
        self.sync()
        n0 = self.net.contents
        b0 = n0.bptt.contents
        S, D, I, H, O = self.S, self.D, self.I, self.H, self.O
        snap = {
            "ih_w": rc.view(n0.ih_weights, I, H).copy(),
            "ho_w": rc.view(n0.ho_weights, H, O).copy(),
            "ih_m": rc.view(b0.ih_momentum, I, H).copy(),
            "ho_m": rc.view(b0.ho_momentum, H, O).copy(),
            "ih_delta": rc.view(b0.ih_delta, I, H).copy(),
            "ho_delta": rc.view(b0.ho_delta, H, O).copy(),
        }
        if self.bottom_inputs:
            bl = n0.bottom_layer.contents
            snap.update(b_w=rc.view(bl.weights, bl.i_size, bl.o_size).copy(),
                        b_m=rc.view(bl.momentums, bl.i_size, bl.o_size).copy(),
                        b_delta=rc.view(bl.delta, bl.i_size, bl.o_size).copy(),
                        b_o_error=rc.view(bl.o_error, bl.o_size).copy())
        hist = np.zeros((D, S, I), np.float32)
        hidden = np.zeros((S, H), np.float32)
        output = np.zeros((S, O), np.float32)
        o_error = np.zeros((S, O), np.float32)
        index = np.zeros(S, np.int32)
        mef = np.zeros(S, np.float32)
        ih_scale = np.zeros(S, np.float32)
        gen = np.zeros(S, np.uint32)
        rng = np.zeros((S, 4), np.uint64)
        for j in range(S):
            n = self.nets[j].contents
            b = n.bptt.contents
            hist[:, j, :] = rc.view(b.history, D, I)
            hidden[j] = rc.view(n.hidden_layer, H)
            output[j] = rc.view(n.output_layer, O)
            o_error[j] = rc.view(b.o_error, O)
            index[j] = b.index
            mef[j] = b.min_error_factor
            ih_scale[j] = b.ih_scale
            gen[j] = n.generation
            rng[j] = (n.rng.a, n.rng.b, n.rng.c, n.rng.d)
        snap.update(hist=hist, hidden=hidden, output=output, o_error=o_error, index=index,
                    min_error_factor=mef, ih_scale=ih_scale, generation=gen, rng=rng)
        return snap


class AmdBatchedSet(ApiSet):
    """librecur_amd.so driven through its additive batched entry points
    (include/recur_amd.h part 2): the whole set per call, text on the device."""

    def __init__(self, lib, *args, **kw):
        super().__init__(lib, *args, **kw)
        self.handle = lib.rnn_amd_set_open(self.nets, self.S)
        assert self.handle, "rnn_amd_set_open failed"
        self._text = None

    def load_text(self, text):
        self._text = np.ascontiguousarray(text, dtype=np.uint8)
        self.lib.rnn_amd_set_load_text(self.handle, rc.u8ptr(self._text), len(self._text))

    def char_step_deltas(self, text, i):
        if self._text is None or self._text is not text:
            self.load_text(text)
        self.lib.rnn_amd_set_char_step_deltas(self.handle, i)

    def char_step(self, text, i, method=rc.WEIGHTED, momentum=0.95):
        if self._text is None or (self._text is not text and not np.array_equal(self._text, text)):
            self.load_text(text)
        self.lib.rnn_amd_set_char_step(self.handle, i, method, momentum)

    def stats(self, clear=False):
        st = rc.AmdStats()
        self.lib.rnn_amd_set_read_stats(self.handle, C.byref(st), int(clear))
        return st

    def close(self):
        self.lib.rnn_amd_set_close(self.handle)
        super().close()
