"""ctypes views of the C libraries the tests talk to.

* ``load_amd()``  -> recur_amd/lib/librecur_amd.so, the product (HIP, gfx950): recur_amd/api.py, re-exported here
* ``load_ref()``  -> oracle/_ref/librecur_ref.so, the real reference compiled from
  /root/reference by oracle/Makefile (absent on machines without the reference
  unless the prebuilt file travelled there)
* ``load_oracle()`` -> oracle/liboracle.so, this repo's CPU restatement

The product and the reference share one ABI (include/recur_amd.h mirrors
recur-nn.h:158-334), so one set of struct definitions and prototypes serves
both, and a test body can be run against either library.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recur_amd.api import *  # noqa: F401,F403  (structs, prototypes, helpers, load_amd)
from recur_amd.api import _bind, ROOT, c_float_p, c_int_p, c_u8_p  # noqa: F401

REF_LIB = os.path.join(ROOT, "oracle", "_ref", "librecur_ref.so")
REF_FAST_LIB = os.path.join(ROOT, "oracle", "_ref", "librecur_ref_fast.so")
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")
ORACLE_FAST_LIB = os.path.join(ROOT, "oracle", "liboracle_fast.so")

REF_MISSING = {"rnn_load_net", "rnn_save_net"}

REF_SHIM_API = {
    "ref_rand64": (C.c_uint64, [C.POINTER(RandCtx)]),
    "ref_init_rand64": (None, [C.POINTER(RandCtx), C.c_uint64]),
    "ref_rand_double": (C.c_double, [C.POINTER(RandCtx)]),
    "ref_rand_small_int": (C.c_int, [C.POINTER(RandCtx), C.c_int]),
    "ref_cheap_gaussian_noise": (C.c_float, [C.POINTER(RandCtx)]),
    "ref_fast_expf": (C.c_float, [C.c_float]),
    "ref_fast_sigmoid": (C.c_float, [C.c_float]),
    "ref_softmax": (None, [c_float_p, c_float_p, C.c_int]),
    "ref_softmax_best_guess": (C.c_int, [c_float_p, c_float_p, C.c_int]),
    "ref_soft_clip": (C.c_float, [C.c_float, C.c_float]),
    "ref_biased_softmax": (None, [c_float_p, c_float_p, C.c_int, C.c_float]),
    "ref_hash32": (C.c_uint32, [C.c_char_p]),
    "ref_write_utf8_char": (C.c_int, [C.c_uint, C.c_char_p]),
    "ref_read_utf8_char": (C.c_int, [C.c_char_p, c_int_p]),
    "ref_sizeof_net": (C.c_int, []),
    "ref_sizeof_bptt": (C.c_int, []),
}


def have_ref():
    return os.path.exists(REF_LIB)


def load_ref(fast=False):
    lib = C.CDLL(REF_FAST_LIB if fast else REF_LIB)
    _bind(lib, RNN_API, skip=REF_MISSING)
    _bind(lib, REF_SHIM_API)
    return lib


# ------------------------------------------------------------------ oracle --

class OrcRng(C.Structure):
    _fields_ = [("a", C.c_uint64), ("b", C.c_uint64), ("c", C.c_uint64), ("d", C.c_uint64)]


class OrcSet(C.Structure):
    _fields_ = [
        ("input_size", C.c_int), ("hidden_size", C.c_int), ("output_size", C.c_int),
        ("I", C.c_int), ("H", C.c_int), ("O", C.c_int),
        ("S", C.c_int), ("D", C.c_int),
        ("activation", C.c_int),
        ("flags", C.c_uint32),
        ("ih_w", c_float_p), ("ho_w", c_float_p),
        ("ih_m", c_float_p), ("ho_m", c_float_p),
        ("ih_aux", c_float_p), ("ho_aux", c_float_p),
        ("ih_delta", c_float_p), ("ho_delta", c_float_p), ("ih_delta_tmp", c_float_p),
        ("hist", c_float_p), ("hidden", c_float_p), ("output", c_float_p),
        ("o_error", c_float_p), ("err_a", c_float_p), ("err_b", c_float_p),
        ("index", c_int_p),
        ("learn_rate", c_float_p),
        ("min_error_factor", c_float_p),
        ("ih_scale", c_float_p),
        ("top_error_raw", c_float_p), ("top_error_scaled", c_float_p), ("bptt_error", c_float_p),
        ("bptt_depth", c_int_p),
        ("generation", C.POINTER(C.c_uint32)),
        ("rng", C.POINTER(OrcRng)),
        ("ho_scale", C.c_float), ("momentum_weight", C.c_float),
        ("presynaptic_noise", C.c_float),
        ("stat_error", C.c_double), ("stat_entropy", C.c_double),
        ("stat_correct", C.c_long), ("stat_count", C.c_long),
        ("stat_depth", C.c_double), ("stat_zeros", C.c_double),
        ("b_in", C.c_int), ("bI", C.c_int), ("bO", C.c_int),
        ("b_w", c_float_p), ("b_m", c_float_p), ("b_aux", c_float_p), ("b_delta", c_float_p),
        ("b_inputs", c_float_p), ("b_outputs", c_float_p), ("b_o_error", c_float_p),
        ("b_learn_rate_scale", C.c_float),
        ("global_first", C.c_int), ("global_count", C.c_int),
    ]


OrcP = C.POINTER(OrcSet)
ORACLE_API = {
    "orc_rand64": (C.c_uint64, [C.POINTER(OrcRng)]),
    "orc_init_rand64": (None, [C.POINTER(OrcRng), C.c_uint64]),
    "orc_rand_double": (C.c_double, [C.POINTER(OrcRng)]),
    "orc_rand_small_int": (C.c_int, [C.POINTER(OrcRng), C.c_int]),
    "orc_cheap_gaussian_noise": (C.c_float, [C.POINTER(OrcRng)]),
    "orc_fast_expf": (C.c_float, [C.c_float]),
    "orc_softmax": (None, [c_float_p, c_float_p, C.c_int]),
    "orc_softmax_best_guess": (C.c_int, [c_float_p, c_float_p, C.c_int]),
    "orc_soft_clip": (C.c_float, [C.c_float, C.c_float]),
    "orc_capped_log2f": (C.c_float, [C.c_float]),
    "orc_momentum_soft_start": (C.c_float, [C.c_float, C.c_float, C.c_float]),
    "orc_padded_sizes": (None, [C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p]),
    "orc_set_new": (OrcP, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32,
                           C.c_float, C.c_uint64]),
    "orc_set_free": (None, [OrcP]),
    "orc_set_add_bottom": (None, [OrcP, C.c_int]),
    "orc_set_seed_clones": (None, [OrcP]),
    "orc_set_init_flat": (None, [OrcP, C.c_float, C.c_int, C.c_double]),
    "orc_advance": (None, [OrcP, C.c_int]),
    "orc_opinion": (c_float_p, [OrcP, C.c_int, c_float_p, C.c_float]),
    "orc_one_hot_opinion": (c_float_p, [OrcP, C.c_int, C.c_int, C.c_float]),
    "orc_net_error_bptt": (C.c_float, [OrcP, C.c_int, C.c_int, C.c_int, c_int_p]),
    "orc_calc_deltas": (None, [OrcP, C.c_int, C.c_int, c_int_p]),
    "orc_grouped_softmax_error": (C.c_int, [OrcP, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p, c_float_p,
                                            c_int_p, c_float_p]),
    "orc_multi_softmax_error": (C.c_float, [OrcP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, c_int_p]),
    "orc_clear_deltas": (None, [OrcP]),
    "orc_apply_learning": (None, [OrcP, C.c_int, C.c_float]),
    "orc_condition": (None, [OrcP, C.c_uint32]),
    "orc_bptt_calculate": (None, [OrcP, C.c_int, C.c_uint, C.c_float]),
    "orc_cross_entropy": (C.c_double, [OrcP, C.c_int, c_u8_p, C.c_int, C.c_int]),
    "orc_multi_cross_entropy": (None, [OrcP, C.c_int, c_u8_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]),
    "orc_fast_sigmoid": (C.c_float, [C.c_float]),
    "orc_sigmoid_mse_error": (None, [OrcP, C.c_int, c_float_p, C.c_int]),
    "orc_multitext_train": (None, [OrcP, C.c_int, c_u8_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                   C.c_float, C.c_int, c_float_p, c_float_p]),
    "orc_set_char_step": (None, [OrcP, c_u8_p, C.c_int, C.c_int, C.c_int, C.c_float]),
    "orc_set_char_step_deltas": (None, [OrcP, c_u8_p, C.c_int, C.c_int]),
}


def build_oracle():
    """Compile oracle/ (gcc only); also refreshes oracle/_ref when the reference
    sources are present.  Building the checker is not using it."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def load_oracle(fast=False):
    path = ORACLE_FAST_LIB if fast else ORACLE_LIB
    if not os.path.exists(path):
        build_oracle()
    return _bind(C.CDLL(path), ORACLE_API)


