"""ctypes view of librecur_amd.so: the struct layouts and prototypes of its C API (include/recur_amd.h mirrors
recur-nn.h:158-334, so the same table fits any library with that ABI), numpy helpers, and two small drivers of the
API that benchmarks, tools and tests share: ``ApiSet`` (a training set through the per-net rnn_* calls) and
``AmdBatchedSet`` (the same through the additive batched calls, text on the device).  No checker lives here: the
oracle and the compiled reference are loaded by tests/recur_ctypes.py.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AMD_LIB = os.path.join(ROOT, "recur_amd", "lib", "librecur_amd.so")


c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_u8_p = C.POINTER(C.c_uint8)


class RandCtx(C.Structure):
    _fields_ = [("a", C.c_uint64), ("b", C.c_uint64), ("c", C.c_uint64), ("d", C.c_uint64)]


class RecurNNBPTT(C.Structure):
    _fields_ = [
        ("depth", C.c_int),
        ("index", C.c_int),
        ("i_error", c_float_p),
        ("h_error", c_float_p),
        ("o_error", c_float_p),
        ("ih_momentum", c_float_p),
        ("ho_momentum", c_float_p),
        ("history", c_float_p),
        ("ih_delta", c_float_p),
        ("ho_delta", c_float_p),
        ("ih_delta_tmp", c_float_p),
        ("ih_aux", c_float_p),
        ("ho_aux", c_float_p),
        ("mem", c_float_p),
        ("learn_rate", C.c_float),
        ("ih_scale", C.c_float),
        ("ho_scale", C.c_float),
        ("momentum", C.c_float),
        ("momentum_weight", C.c_float),
        ("min_error_factor", C.c_float),
    ]


class RecurExtraLayer(C.Structure):
    _fields_ = [
        ("mem", c_float_p),
        ("weights", c_float_p),
        ("momentums", c_float_p),
        ("aux", c_float_p),
        ("delta", c_float_p),
        ("inputs", c_float_p),
        ("outputs", c_float_p),
        ("i_error", c_float_p),
        ("o_error", c_float_p),
        ("learn_rate_scale", C.c_float),
        ("input_size", C.c_int),
        ("output_size", C.c_int),
        ("i_size", C.c_int),
        ("o_size", C.c_int),
        ("overlap", C.c_int),
    ]


class RecurNN(C.Structure):
    _fields_ = [
        ("i_size", C.c_int),
        ("h_size", C.c_int),
        ("o_size", C.c_int),
        ("input_size", C.c_int),
        ("hidden_size", C.c_int),
        ("output_size", C.c_int),
        ("ih_size", C.c_int),
        ("ho_size", C.c_int),
        ("flags", C.c_uint32),
        ("log", C.c_void_p),
        ("mem", c_float_p),
        ("input_layer", c_float_p),
        ("hidden_layer", c_float_p),
        ("output_layer", c_float_p),
        ("ih_weights", c_float_p),
        ("ho_weights", c_float_p),
        ("real_inputs", c_float_p),
        ("rng", RandCtx),
        ("bptt", C.POINTER(RecurNNBPTT)),
        ("bottom_layer", C.POINTER(RecurExtraLayer)),
        ("metadata", C.c_char_p),
        ("generation", C.c_uint32),
        ("presynaptic_noise", C.c_float),
        ("activation", C.c_int),
    ]


class InitParams(C.Structure):
    _fields_ = [
        ("method", C.c_int),
        ("submethod", C.c_int),
        ("bias_uses_submethod", C.c_int),
        ("inputs_use_submethod", C.c_int),
        ("fan_in_sum", C.c_float),
        ("fan_in_step", C.c_float),
        ("fan_in_min", C.c_float),
        ("fan_in_ratio", C.c_float),
        ("flat_variance", C.c_float),
        ("flat_shape", C.c_int),
        ("flat_perforation", C.c_double),
        ("run_input_probability", C.c_float),
        ("run_input_magnitude", C.c_float),
        ("run_gain", C.c_float),
        ("run_len_mean", C.c_float),
        ("run_len_stddev", C.c_float),
        ("run_n", C.c_int),
        ("run_loop", C.c_int),
        ("run_crossing_paths", C.c_int),
        ("run_inputs_miss", C.c_int),
        ("run_input_at_start", C.c_int),
    ]


class ErrorRange(C.Structure):
    _fields_ = [("start", C.c_int), ("len", C.c_int)]


class AmdStats(C.Structure):
    _fields_ = [
        ("error", C.c_double),
        ("entropy", C.c_double),
        ("correct", C.c_long),
        ("count", C.c_long),
        ("bptt_depth_sum", C.c_double),
        ("hidden_zeros", C.c_double),
    ]


NetP = C.POINTER(RecurNN)

# flags / enums (include/recur_amd.h)
FLAG_OWN_BPTT = 1
FLAG_OWN_WEIGHTS = 2
FLAG_LOG_HIDDEN_SUM = 16
FLAG_ADAPTIVE_MIN_ERROR = 64
FLAG_NO_MOMENTUMS = 128
FLAG_NO_DELTAS = 256
FLAG_BOTTOM_LAYER = 1024
FLAG_AUX_ARRAYS = 2048
COND_USE_SCALE = 1 << 16
COND_USE_ZERO = 1 << 18
COND_USE_LAWN_MOWER = 1 << 19
COND_USE_TALL_POPPY = 1 << 20
COND_USE_RAND = 1 << 22
FLAG_STANDARD = FLAG_OWN_BPTT | FLAG_OWN_WEIGHTS | COND_USE_ZERO | FLAG_LOG_HIDDEN_SUM
RELU, RESQRT, RECLIP20 = 1, 2, 5
WEIGHTED, NESTEROV, SIMPLIFIED_NESTEROV, CLASSICAL, ADAGRAD, ADADELTA, RPROP = range(7)
INIT_ZERO, INIT_FLAT, INIT_FAN_IN, INIT_RUNS = range(4)
DIST_UNIFORM, DIST_GAUSSIAN, DIST_LOG_NORMAL, DIST_SEMICIRCLE = 1, 2, 3, 4
SUBSEED = C.c_uint64(-2).value

# The drop-in surface: name -> (restype, argtypes).  This is recur-nn.h:269-334.
RNN_API = {
    "rnn_new": (NetP, [C.c_uint, C.c_uint, C.c_uint, C.c_uint32, C.c_uint64, C.c_char_p,
                       C.c_int, C.c_float, C.c_float, C.c_float, C.c_int]),
    "rnn_clone": (NetP, [NetP, C.c_uint32, C.c_uint64, C.c_char_p]),
    "rnn_new_extra_layer": (C.POINTER(RecurExtraLayer), [C.c_int, C.c_int, C.c_int, C.c_uint32]),
    "rnn_new_with_bottom_layer": (NetP, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32,
                                         C.c_uint64, C.c_char_p, C.c_int, C.c_float,
                                         C.c_float, C.c_float, C.c_int, C.c_int]),
    "rnn_set_log_file": (None, [NetP, C.c_char_p, C.c_int]),
    "rnn_randomise_weights_clever": (None, [NetP, C.POINTER(InitParams)]),
    "rnn_randomise_weights_simple": (None, [NetP, C.c_int]),
    "rnn_randomise_weights_auto": (None, [NetP]),
    "rnn_init_default_weight_parameters": (None, [NetP, C.POINTER(InitParams)]),
    "rnn_scale_initial_weights": (None, [NetP, C.c_float]),
    "rnn_print_net_stats": (None, [NetP]),
    "rnn_delete_net": (None, [NetP]),
    "rnn_new_training_set": (C.POINTER(NetP), [NetP, C.c_int]),
    "rnn_delete_training_set": (None, [C.POINTER(NetP), C.c_int, C.c_int]),
    "rnn_opinion": (c_float_p, [NetP, c_float_p, C.c_float]),
    "rnn_multi_pgm_dump": (None, [NetP, C.c_char_p, C.c_char_p]),
    "rnn_load_net": (NetP, [C.c_char_p]),
    "rnn_save_net": (C.c_int, [NetP, C.c_char_p, C.c_int]),
    "rnn_bptt_clear_deltas": (None, [NetP]),
    "rnn_bptt_advance": (None, [NetP]),
    "rnn_bptt_calculate": (None, [NetP, C.c_uint]),
    "rnn_apply_learning": (None, [NetP, C.c_int, C.c_float]),
    "rnn_calculate_momentum_soft_start": (C.c_float, [C.c_float, C.c_float, C.c_float]),
    "rnn_bptt_calc_deltas": (None, [NetP, C.c_int, C.POINTER(ErrorRange)]),
    "rnn_condition_net": (None, [NetP]),
    "rnn_log_net": (None, [NetP]),
    "rnn_forget_history": (None, [NetP, C.c_int]),
    "rnn_perforate_weights": (None, [NetP, C.c_float]),
    "rnn_weight_noise": (None, [NetP, C.c_float]),
    "rnn_set_momentum_values": (None, [NetP, C.c_float]),
    "rnn_set_aux_values": (None, [NetP, C.c_float]),
    "rnn_zap_non_diagonals": (None, [NetP, C.c_int, C.c_int, C.c_int]),
    "rnn_clear_diagonal_only_section": (None, [NetP, C.c_uint, C.c_uint]),
}
# recur-nn-io.c needs tinycdb, so the compiled reference lacks these two.

# Additive entry points of include/recur_amd.h part 2.
AMD_API = {
    "rnn_amd_device_count": (C.c_int, []),
    "rnn_amd_use_device": (None, [C.c_int, C.c_void_p]),
    "rnn_amd_current_stream": (C.c_void_p, []),
    "rnn_amd_version": (C.c_char_p, []),
    "rnn_amd_sync_host": (None, [NetP, C.c_int]),
    "rnn_amd_host_written": (None, [NetP, C.c_int]),
    "rnn_amd_set_open": (C.c_void_p, [C.POINTER(NetP), C.c_int]),
    "rnn_amd_set_close": (None, [C.c_void_p]),
    "rnn_amd_set_drop": (None, [C.c_void_p]),
    "rnn_amd_set_size": (C.c_int, [C.c_void_p]),
    "rnn_amd_set_advance": (None, [C.c_void_p]),
    "rnn_amd_set_opinion": (None, [C.c_void_p, c_float_p, C.c_int, c_float_p]),
    "rnn_amd_set_one_hot_opinion": (None, [C.c_void_p, c_int_p, c_float_p]),
    "rnn_amd_set_put_o_error": (None, [C.c_void_p, c_float_p, C.c_int]),
    "rnn_amd_set_softmax_error": (None, [C.c_void_p, c_int_p]),
    "rnn_amd_set_calc_deltas": (None, [C.c_void_p, C.c_int, C.POINTER(ErrorRange), c_u8_p]),
    "rnn_amd_set_load_text": (None, [C.c_void_p, c_u8_p, C.c_int]),
    "rnn_amd_set_char_step": (None, [C.c_void_p, C.c_int, C.c_int, C.c_float]),
    "rnn_amd_set_exchange_export": (None, [C.c_void_p, C.c_void_p]),
    "rnn_amd_set_exchange_join": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "rnn_amd_set_exchange_leave": (None, [C.c_void_p]),
    "rnn_amd_set_apply_exchange": (None, [C.c_void_p, C.c_int, C.c_float]),
    "rnn_amd_set_exchange_range": (None, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "rnn_amd_set_replica_checksum": (C.c_uint64, [C.c_void_p, C.c_int, C.c_int]),
    "rnn_amd_set_char_step_fused": (None, [C.c_void_p, C.c_int, C.c_uint]),
    "rnn_amd_set_read_stats": (None, [C.c_void_p, C.POINTER(AmdStats), C.c_int]),
    "rnn_amd_set_external_delta": (None, [C.c_void_p, C.c_void_p]),
    "rnn_amd_set_char_step_deltas": (None, [C.c_void_p, C.c_int]),
    "rnn_amd_set_shard": (None, [C.c_void_p, C.c_int, C.c_int]),
    "rnn_amd_run_text": (C.c_double, [NetP, c_u8_p, C.c_int, C.c_int]),
    "rnn_amd_set_grouped_softmax_error": (None, [C.c_void_p, C.c_int, c_int_p, c_int_p, c_int_p, c_float_p,
                                                 c_u8_p]),
    "rnn_amd_set_multi_step_deltas": (None, [C.c_void_p, c_int_p, c_int_p, c_int_p, C.c_int, C.c_float, C.c_int]),
    "rnn_amd_set_multi_step": (None, [C.c_void_p, c_int_p, c_int_p, c_int_p, C.c_int, C.c_float, C.c_int, C.c_float]),
    "rnn_amd_set_multi_text_loss": (None, [C.c_void_p, C.c_int, c_int_p, C.c_int, C.c_float]),
    "rnn_amd_set_multi_calc_deltas": (None, [C.c_void_p, C.c_int]),
    "rnn_amd_set_text_opinion": (None, [C.c_void_p, C.c_int, C.c_int]),
    "rnn_amd_set_sigmoid_mse_error": (None, [C.c_void_p, c_float_p, C.c_int, C.c_int]),
    "rnn_amd_set_sigmoid_outputs": (None, [C.c_void_p, C.c_int, c_float_p]),
    "rnn_amd_set_opinion_sigmoid_mse": (None, [C.c_void_p, c_float_p, C.c_int, c_float_p, C.c_int, C.c_int]),
    "rnn_amd_set_dense_step_sigmoid_mse": (None, [C.c_void_p, c_float_p, C.c_int, c_float_p, C.c_int, C.c_int, C.c_int,
                                                  C.c_float]),
    "rnn_amd_set_opinion_grouped_softmax": (None, [C.c_void_p, c_float_p, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p,
                                                   c_float_p, c_u8_p]),
    "rnn_amd_run_text_heads": (None, [NetP, c_u8_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "rnn_amd_synchronize": (None, []),
    "rnn_amd_dist_get_id": (C.c_int, [C.c_void_p]),
    "rnn_amd_dist_init": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "rnn_amd_dist_finalize": (None, []),
    "rnn_amd_dist_rank": (C.c_int, []),
    "rnn_amd_dist_world": (C.c_int, []),
    "rnn_amd_dist_all_reduce": (None, [C.c_void_p, C.c_size_t]),
    "rnn_amd_dist_max": (C.c_double, [C.c_double]),
    "rnn_amd_dist_barrier": (None, []),
    "rnn_amd_new_training_set_shard": (C.POINTER(NetP), [NetP, C.c_int, C.c_int, C.c_int]),
    "rnn_amd_set_dist_all_reduce_deltas": (None, [C.c_void_p]),
    "rnn_amd_kernel_time_enable": (None, [C.c_int]),
    "rnn_amd_kernel_time_ms": (C.c_double, [C.c_int, C.POINTER(C.c_long), C.c_int]),
}
RNN_AMD_WEIGHTS, RNN_AMD_MOMENTUMS, RNN_AMD_DELTAS, RNN_AMD_STREAM, RNN_AMD_ALL_STREAMS = 1, 2, 4, 8, 16
RNN_AMD_EVERYTHING = 31
RNN_AMD_EXCHANGE_BLOB_BYTES = 256

def _bind(lib, table, skip=()):
    for name, (res, args) in table.items():
        if name in skip:
            continue
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def load_amd():
    if not os.path.exists(AMD_LIB):
        raise RuntimeError("librecur_amd.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(os.environ.get("RECUR_AMD_LIB", AMD_LIB))  # development: an experimental build
    _bind(lib, RNN_API)
    _bind(lib, AMD_API)
    return lib


# ------------------------------------------------------------ numpy helpers --

def view(ptr, *shape):
    """numpy view (no copy) of the C float/int array behind a ctypes pointer."""
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype=np.float32)
    return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(shape)


def fptr(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_float_p)


def iptr(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_int_p)


def u8ptr(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_u8_p)


def rel_err(a, b):
    """|a-b| / |b| in the 2-norm: the parity metric for fp32 arrays."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d = np.linalg.norm(a - b)
    n = np.linalg.norm(b)
    return d / n if n > 0 else d


def max_err(a, b):
    """max |a-b| / max |b|: the element-wise companion of rel_err."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    d = np.abs(a - b).max()
    n = np.abs(b).max()
    return d / n if n > 0 else d


def elem_err(a, b, floor=1e-2):
    """max over the elements that are not small (|b| >= floor * max|b|) of |a-b| / |b|: "relative" read element by
    element.  floor 1e-2: between the reference's own strict and -Ofast builds, one generation deep from identical
    state, this reads <= 2.7e-5 on every array, while with floor 1e-3 it reaches 3.9e-4 on ih_delta at hidden 1024 (a
    sum of K products carries ~sqrt(K) eps of its largest partial sums, whatever its own size):
    profiles/r05_reference_elementwise_self_difference.txt."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    top = np.abs(b).max()
    if not top > 0:
        return float(np.abs(a - b).max())
    big = np.abs(b) >= floor * top
    return float((np.abs(a - b)[big] / np.abs(b)[big]).max())


# ---------------------------------------------------------- character model --

class CharAlphabet(C.Structure):
    _fields_ = [("points", c_int_p), ("collapsed_points", c_int_p), ("len", C.c_int),
                ("collapsed_len", C.c_int), ("flags", C.c_uint32)]


AlphaP = C.POINTER(CharAlphabet)


class CharModel(C.Structure):
    pass


class CharSchedule(C.Structure):  # charmodel.h:26-34
    _fields_ = [("recent", c_float_p), ("recent_len", C.c_int), ("timeout", C.c_int),
                ("learn_rate_mul", C.c_float), ("learn_rate_min", C.c_float), ("adjust_noise", C.c_int),
                ("eval", C.CFUNCTYPE(None, C.POINTER(CharModel), C.c_float, C.c_int))]


class CharImageSettings(C.Structure):  # charmodel.h:18-24
    _fields_ = [("basename", C.c_char_p), ("temporal_pgm_dump", C.c_bool), ("input_ppm", C.c_void_p),
                ("error_ppm", C.c_void_p), ("periodic_pgm_dump_string", C.c_char_p)]


CharModel._fields_ = [  # charmodel.h:56-73
    ("net", NetP), ("training_nets", C.POINTER(NetP)), ("n_training_nets", C.c_int),
    ("batch_size", C.c_uint), ("filename", C.c_char_p), ("momentum", C.c_float),
    ("momentum_soft_start", C.c_float), ("learning_style", C.c_int),
    ("periodic_weight_noise", C.c_float), ("report_interval", C.c_uint), ("save_net", C.c_bool),
    ("use_multi_tap_path", C.c_bool), ("alphabet", AlphaP), ("schedule", CharSchedule),
    ("images", CharImageSettings)]


class CharVentropy(C.Structure):  # charmodel.h:36-45
    _fields_ = [("net", NetP), ("counter", C.c_int), ("history", c_float_p), ("text", c_u8_p),
                ("len", C.c_int), ("lap", C.c_int), ("lapsize", C.c_int), ("entropy", C.c_float)]


class CharMetadata(C.Structure):  # charmodel.h:75-81
    _fields_ = [("alphabet", C.c_char_p), ("collapse_chars", C.c_char_p), ("utf8", C.c_bool),
                ("case_insensitive", C.c_bool), ("collapse_space", C.c_bool)]


MetaP = C.POINTER(CharMetadata)
class CharProgressReport(C.Structure):  # charmodel.h:132-138
    _fields_ = [("training_entropy", C.c_float), ("training_error", C.c_float),
                ("training_accuracy", C.c_float), ("per_second", C.c_float)]


class CharMultiConfab(C.Structure):  # charmodel.h:140-152
    _fields_ = [("nets", C.POINTER(NetP)), ("last_char", c_int_p), ("caps_marker", C.c_int),
                ("strings", C.POINTER(C.c_char_p)), ("n_classes", C.c_uint), ("char_len", C.c_uint),
                ("byte_len", C.c_uint), ("bias", C.c_float), ("period", C.c_uint), ("alphabet", AlphaP)]


CHAR_API = {
    "rnn_char_multitext_train": (None, [NetP, c_u8_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                        C.POINTER(CharProgressReport), C.POINTER(CharMultiConfab), C.c_int,
                                        C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]),
    "rnn_char_multitext_spin": (None, [NetP, c_u8_p, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_int]),
    "rnn_char_multi_cross_entropy": (None, [NetP, c_u8_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]),
    "rnn_char_new_multi_confab": (C.POINTER(CharMultiConfab), [NetP, AlphaP, C.c_int, C.c_int, C.c_uint, C.c_int]),
    "rnn_char_free_multi_confab": (None, [C.POINTER(CharMultiConfab)]),
    "rnn_char_init_schedule": (None, [C.POINTER(CharSchedule), C.c_int, C.c_float, C.c_float, C.c_int]),
    "rnn_char_calc_ventropy": (C.c_float, [C.POINTER(CharModel), C.POINTER(CharVentropy), C.c_int]),
    "rnn_char_delete_ventropy": (None, [C.POINTER(CharVentropy)]),
    "rnn_char_init_ventropy": (None, [C.POINTER(CharVentropy), NetP, c_u8_p, C.c_int, C.c_int]),
    "rnn_char_confabulate": (C.c_int, [NetP, C.c_char_p, C.c_int, C.c_int, AlphaP, C.c_float, c_int_p,
                                       C.c_int, C.c_int]),
    "rnn_char_epoch": (C.c_int, [C.POINTER(CharModel), NetP, C.POINTER(CharVentropy), c_u8_p, C.c_int,
                                 C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_uint,
                                 C.c_uint]),
    "rnn_char_prime": (C.c_int, [NetP, AlphaP, c_u8_p, C.c_int]),
    "rnn_char_cross_entropy": (C.c_double, [NetP, AlphaP, c_u8_p, C.c_int, C.c_int, c_u8_p, C.c_int]),
    "rnn_char_uncollapse_text": (C.c_void_p, [AlphaP, c_u8_p, C.c_int, c_int_p]),
    "rnn_char_dump_collapsed_text": (None, [c_u8_p, C.c_int, C.c_char_p, C.c_char_p]),
    "rnn_char_construct_metadata": (C.c_void_p, [MetaP]),
    "rnn_char_load_metadata": (C.c_int, [C.c_char_p, MetaP]),
    "rnn_char_free_metadata_items": (None, [MetaP]),
    "rnn_char_construct_net_filename": (C.c_void_p, [MetaP, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rnn_char_check_metadata": (C.c_int, [NetP, MetaP, C.c_bool, C.c_bool]),
    "rnn_char_copy_metadata_items": (None, [MetaP, MetaP]),
    "rnn_char_dump_alphabet": (None, [AlphaP]),
    "rnn_char_get_codepoint": (C.c_int, [AlphaP, C.c_char_p]),
    "rnn_char_new_alphabet_from_net": (AlphaP, [NetP]),
    "rnn_char_new_alphabet": (AlphaP, []),
    "rnn_char_free_alphabet": (None, [AlphaP]),
    "rnn_char_alphabet_set_flags": (None, [AlphaP, C.c_bool, C.c_bool, C.c_bool]),
    "rnn_char_find_alphabet_s": (C.c_int, [C.c_char_p, C.c_int, AlphaP, C.c_double, C.c_double, C.c_double]),
    "rnn_char_find_alphabet_f": (C.c_int, [C.c_char_p, AlphaP, C.c_double, C.c_double, C.c_double]),
    "rnn_char_new_char_lut": (c_int_p, [AlphaP]),
    "rnn_char_alloc_encoded_text": (c_u8_p, [AlphaP, C.c_char_p, C.c_int, c_int_p, c_int_p, C.c_bool]),
    "rnn_char_load_new_encoded_text": (c_u8_p, [C.c_char_p, AlphaP, c_int_p, C.c_int]),
    "rnn_char_alloc_file_contents": (C.c_int, [C.c_char_p, C.POINTER(C.c_char_p), c_int_p]),
}


class ClassifyMetadata(C.Structure):  # gstclassify.h:57-72 / include/recur_amd_classify.h
    _fields_ = [("classes", C.c_char_p), ("min_freq", C.c_float), ("max_freq", C.c_float), ("knee_freq", C.c_float),
                ("mfccs", C.c_int), ("window_size", C.c_int), ("basename", C.c_char_p), ("delta_features", C.c_int),
                ("focus_freq", C.c_float), ("lag", C.c_float), ("intensity_feature", C.c_int),
                ("confirmation_lag", C.c_float), ("features_offset", C.c_char_p), ("features_scale", C.c_char_p)]


class BalancedTraining(C.Structure):
    _fields_ = [("n_outputs", C.c_int), ("bias", C.c_float), ("seen", C.POINTER(C.c_uint32)),
                ("used", C.POINTER(C.c_uint32)), ("train_p", c_float_p)]


CLASSIFY_API = {
    "rnn_amd_classify_construct_metadata": (C.c_void_p, [C.POINTER(ClassifyMetadata)]),
    "rnn_amd_classify_load_metadata": (C.c_int, [C.c_char_p, C.POINTER(ClassifyMetadata)]),
    "rnn_amd_classify_free_metadata_items": (None, [C.POINTER(ClassifyMetadata)]),
    "rnn_amd_classify_net_filename": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   C.c_int]),
    "rnn_amd_classify_parse_classes": (C.c_int, [C.c_char_p, c_int_p, c_int_p, C.c_int, c_int_p, c_int_p]),
    "rnn_amd_classify_check_net": (C.c_int, [NetP, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rnn_amd_balanced_new": (C.POINTER(BalancedTraining), [C.c_int, C.c_float]),
    "rnn_amd_balanced_free": (None, [C.POINTER(BalancedTraining)]),
    "rnn_amd_balanced_begin": (None, [C.POINTER(BalancedTraining)]),
    "rnn_amd_classify_generation": (C.c_int, [C.c_void_p, c_float_p, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p,
                                              c_float_p, C.POINTER(BalancedTraining), C.c_int, C.c_float, C.c_int]),
}


def bind_classify(lib):
    return _bind(lib, CLASSIFY_API)


CHAR_CASE_INSENSITIVE, CHAR_UTF8, CHAR_COLLAPSE_SPACE = 1, 2, 4
# text-predict's default character set (text-predict.c:44-45)
DEFAULT_CHARSET = b"8 etaonihsrdlucmwfygpb,v.k-;x\"qj'?:z)(_!*&"
DEFAULT_COLLAPSE_CHARS = b"10872}{659/34][@"
EREWHON = os.path.join(ROOT, "tests", "golden", "erewhon.txt")


def bind_char(lib):
    return _bind(lib, CHAR_API)


def default_text_alphabet(lib):
    """The alphabet text-predict builds from its defaults (text-predict.c:698-719):
    case-insensitive, bytes, collapsed space; note that it fills collapsed_points
    from the ALPHABET string, which leaves the collapse characters
    to the table's default, space."""
    a = lib.rnn_char_new_alphabet()
    lib.rnn_char_alphabet_set_flags(a, True, False, True)
    for i, c in enumerate(DEFAULT_CHARSET):
        a.contents.points[i] = c
        a.contents.collapsed_points[i] = c
    a.contents.len = len(DEFAULT_CHARSET)
    a.contents.collapsed_len = len(DEFAULT_CHARSET)
    return a


def encode_erewhon(lib):
    """erewhon.txt as the u8 symbol stream text-predict trains on."""
    bind_char(lib)
    a = default_text_alphabet(lib)
    n = C.c_int(0)
    p = lib.rnn_char_load_new_encoded_text(EREWHON.encode(), a, C.byref(n), 2)
    out = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
    lib.rnn_char_free_alphabet(a)
    return out
