"""Scenario drivers shared by the parity tests.

``ApiSet`` drives a library that speaks the recur-nn.h ABI (the compiled
reference, or librecur_amd.so through its per-net drop-in calls) exactly the way
the reference's own caller does (rnn_char_epoch, charmodel-predict.c:288-311).
``OracleSet`` drives oracle/liboracle.so through the same steps.  Both expose
``snapshot()`` in the oracle/device layout so arrays can be compared directly.
"""
import ctypes as C

import numpy as np

import recur_ctypes as rc


_TEXT_CACHE = {}


def synthetic_text(n, alphabet=42, seed=7):
    """Seeded symbol stream: SURVEY.md section 8(d)'s data-free variant (symbols from
    rand_small_int(seed 7, 42)), from the numpy restatement of the generator: input data
    does not come out of the checker's library."""
    import golden_cases as gc
    key = (alphabet, seed)
    if key not in _TEXT_CACHE or len(_TEXT_CACHE[key]) < n:
        _TEXT_CACHE[key] = gc.synthetic_text_np(max(n, 8000), alphabet, seed)
    return _TEXT_CACHE[key][:n].copy()


def synthetic_text_oracle(n, alphabet=42, seed=7):
    """The same stream from the oracle's generator (cross-check of the two)."""
    orc = rc.load_oracle()
    rng = rc.OrcRng()
    orc.orc_init_rand64(C.byref(rng), seed)
    out = np.empty(n, dtype=np.uint8)
    for i in range(n):
        out[i] = orc.orc_rand_small_int(C.byref(rng), alphabet)
    return out


from recur_amd.drivers import ApiSet, AmdBatchedSet  # noqa: E402,F401  (the drivers live with the package)


class OracleSet:
    """The same scenario on oracle/liboracle.so."""

    def __init__(self, input_size, hidden_size, output_size, S, D, activation=rc.RELU,
                 flags=rc.FLAG_STANDARD | rc.FLAG_ADAPTIVE_MIN_ERROR, learn_rate=1e-3, seed=1,
                 momentum=0.95, variance=None, shape=rc.DIST_SEMICIRCLE, perforation=0.0,
                 fast=False, noise=0.0, bottom_inputs=0, bottom_rate_scale=1.0):
        self.orc = orc = rc.load_oracle(fast=fast)
        self.z = orc.orc_set_new(input_size, hidden_size, output_size, S, D, activation, flags,
                                 learn_rate, seed)
        self.bottom_inputs = bottom_inputs
        if bottom_inputs:
            orc.orc_set_add_bottom(self.z, bottom_inputs)
            self.z.contents.b_learn_rate_scale = bottom_rate_scale
        z = self.z.contents
        self.S, self.D, self.I, self.H, self.O = S, D, z.I, z.H, z.O
        self.input_size, self.hidden_size, self.output_size = input_size, hidden_size, output_size
        if variance is None:
            variance = np.float32(2.0) / np.float32(z.H)
        orc.orc_set_init_flat(self.z, variance, shape, perforation)
        orc.orc_set_seed_clones(self.z)
        self.z.contents.presynaptic_noise = noise

    def close(self):
        self.orc.orc_set_free(self.z)

    def char_step_deltas(self, text, i):
        self.orc.orc_set_char_step_deltas(self.z, rc.u8ptr(text), len(text), i)

    def char_step(self, text, i, method=rc.WEIGHTED, momentum=0.95):
        self.orc.orc_set_char_step(self.z, rc.u8ptr(text), len(text), i, method, momentum)

    def arrays(self):
        """numpy VIEWS of the oracle's arrays (writable)."""
        z = self.z.contents
        S, D, I, H, O = self.S, self.D, self.I, self.H, self.O
        bottom = {}
        if self.bottom_inputs:
            bottom = {"b_w": rc.view(z.b_w, z.bI, z.bO), "b_m": rc.view(z.b_m, z.bI, z.bO),
                      "b_aux": rc.view(z.b_aux, z.bI, z.bO),
                      "b_delta": rc.view(z.b_delta, z.bI, z.bO),
                      "b_o_error": rc.view(z.b_o_error, z.bO)}
        return {
            **bottom,
            "ih_w": rc.view(z.ih_w, I, H), "ho_w": rc.view(z.ho_w, H, O),
            "ih_m": rc.view(z.ih_m, I, H), "ho_m": rc.view(z.ho_m, H, O),
            "ih_aux": rc.view(z.ih_aux, I, H), "ho_aux": rc.view(z.ho_aux, H, O),
            "ih_delta": rc.view(z.ih_delta, I, H), "ho_delta": rc.view(z.ho_delta, H, O),
            "hist": rc.view(z.hist, D, S, I), "hidden": rc.view(z.hidden, S, H),
            "output": rc.view(z.output, S, O), "o_error": rc.view(z.o_error, S, O),
            "err_a": rc.view(z.err_a, S, I), "err_b": rc.view(z.err_b, S, I),
            "index": np.ctypeslib.as_array(z.index, shape=(S,)),
            "min_error_factor": rc.view(z.min_error_factor, S),
            "learn_rate": rc.view(z.learn_rate, S),
            "ih_scale": rc.view(z.ih_scale, S),
            "top_error_raw": rc.view(z.top_error_raw, S),
            "top_error_scaled": rc.view(z.top_error_scaled, S),
            "bptt_error": rc.view(z.bptt_error, S),
            "bptt_depth": np.ctypeslib.as_array(z.bptt_depth, shape=(S,)),
            "generation": np.ctypeslib.as_array(z.generation, shape=(S,)),
        }

    def snapshot(self):
        a = self.arrays()
        snap = {k: np.array(v, copy=True) for k, v in a.items()}
        z = self.z.contents
        rng = np.zeros((self.S, 4), np.uint64)
        for j in range(self.S):
            r = z.rng[j]
            rng[j] = (r.a, r.b, r.c, r.d)
        snap["rng"] = rng
        return snap


FLOAT_KEYS = ["ih_w", "ho_w", "ih_m", "ho_m", "ih_delta", "ho_delta", "hist", "hidden", "output",
              "o_error", "min_error_factor", "ih_scale", "b_w", "b_m", "b_delta", "b_o_error"]
EXACT_KEYS = ["index", "generation", "rng"]


def compare(got, want, rtol, keys=FLOAT_KEYS, exact=EXACT_KEYS):
    """Returns a list of human-readable mismatches (empty == parity)."""
    bad = []
    for k in keys:
        if k in got and k in want:
            e = rc.rel_err(got[k], want[k])
            if not e <= rtol:
                bad.append("%s: rel err %.3g > %.1g" % (k, e, rtol))
            m = rc.max_err(got[k], want[k])
            if not m <= rtol:
                bad.append("%s: max element err %.3g of max|ref| > %.1g" % (k, m, rtol))
    for k in exact:
        if k in got and k in want and not np.array_equal(got[k], want[k]):
            bad.append("%s: not bit-exact" % k)
    return bad
