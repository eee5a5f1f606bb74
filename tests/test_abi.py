"""The drop-in boundary without a GPU: librecur_amd.so loads, exports every
function include/recur_amd.h declares, lays its structs out like the reference,
and its host-only half (construction, weight initialisation, CDB files) behaves
like the reference.  No compute entry point is called here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import recur_ctypes as rc
import replay

ROOT = rc.ROOT
AMD = rc.load_amd()
Z = replay.golden()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "recur_amd.h")).read()
    src += open(os.path.join(ROOT, "include", "recur_amd_char.h")).read()
    src += open(os.path.join(ROOT, "include", "recur_amd_classify.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(rnn_[a-z0-9_]+)\s*\(", src))
    return sorted(n for n in names if n not in ("rnn_log_float", "rnn_log_int"))  # static inline


def test_every_declared_symbol_is_exported():
    missing = [n for n in declared_functions() if not hasattr(AMD, n)]
    assert missing == []
    assert len(declared_functions()) >= 50
    # and the binding tables of the tests cover the whole header
    bound = set(rc.RNN_API) | set(rc.AMD_API) | set(rc.CHAR_API) | set(rc.CLASSIFY_API)
    assert set(declared_functions()) <= bound


def test_nothing_but_the_api_is_exported():
    """The drop-in is loaded into somebody else's process (a GStreamer element, text-predict): its dynamic symbol table
    holds the `rnn_` names of include/*.h and nothing else -- no kernel host stubs, no ramd_* internals, no cdb container
    functions (recur_amd/csrc/librecur_amd.map)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "recur_amd", "lib", "librecur_amd.so")], capture_output=True, text=True, check=True).stdout
    names = [l.split()[-1] for l in out.splitlines() if l.strip()]
    stray = [n for n in names if not n.startswith("rnn_")]
    assert stray == [], stray[:20]
    assert set(declared_functions()) <= set(names)


def test_struct_layout_is_the_reference_abi():
    # offsets the reference's own Python extension relies on (py-recur-text.c:594-599)
    assert C.sizeof(rc.RecurNN) == 176 and C.sizeof(rc.RecurNNBPTT) == 128
    assert rc.RecurNN.presynaptic_noise.offset == 164
    assert rc.RecurNNBPTT.learn_rate.offset == 104
    assert rc.RecurNNBPTT.ih_scale.offset == 108
    assert rc.RecurNNBPTT.ho_scale.offset == 112
    assert rc.RecurNNBPTT.momentum_weight.offset == 120


def test_header_compiles_as_c_and_cxx(tmp_path):
    prog = '#include "recur-nn.h"\nint main(void){RecurNN n; (void)n; return sizeof(RecurNNBPTT) == 128 ? 0 : 1;}\n'
    for cc, ext, std in (("gcc", "c", "-std=gnu11"), ("g++", "cpp", "-std=c++17")):
        f = tmp_path / ("t." + ext)
        f.write_text(prog)
        exe = tmp_path / ("t_" + ext)
        subprocess.run([cc, std, "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(f), "-o",
                        str(exe)], check=True)
        assert subprocess.run([str(exe)]).returncode == 0


def test_padded_sizes_and_first_ring_slot():
    net = AMD.rnn_new(42, 1024, 42, rc.FLAG_STANDARD, 1, None, 20, 1e-5, 0.95, 0.0, rc.RELU)
    n = net.contents
    assert (n.i_size, n.h_size, n.o_size, n.ih_size, n.ho_size) == (1068, 1028, 44, 1097904, 45232)
    b = n.bptt.contents
    assert b.index == 1 and b.depth == 20                    # recur-nn-init.c:133
    assert C.addressof(n.input_layer.contents) == C.addressof(b.history.contents) + 4 * n.i_size
    assert C.addressof(n.real_inputs.contents) == C.addressof(n.input_layer.contents) + 4 * 1025
    assert b.min_error_factor == np.float32(1e-12) * np.float32(1028)
    assert b.ho_scale == 1.0 and b.momentum_weight == 0.5
    AMD.rnn_delete_net(net)


@pytest.mark.parametrize("name,hidden,shape,perf,seed", [
    ("init_semicircle_h99", 99, rc.DIST_SEMICIRCLE, 0.0, 1),
    ("init_uniform_perf_h99", 99, rc.DIST_UNIFORM, 0.7, 1),
    ("init_gaussian_h1024", 1024, rc.DIST_GAUSSIAN, 0.0, 2),
    ("init_lognormal_h64", 64, rc.DIST_LOG_NORMAL, 0.2, 3),
])
def test_weight_init_bit_exact_with_reference(name, hidden, shape, perf, seed):
    net = AMD.rnn_new(42, hidden, 42, rc.FLAG_STANDARD, seed, None, 10, 1e-3, 0.9, 0.0, rc.RELU)
    p = rc.InitParams()
    AMD.rnn_init_default_weight_parameters(net, C.byref(p))
    p.method, p.flat_shape, p.flat_perforation = rc.INIT_FLAT, shape, perf
    AMD.rnn_randomise_weights_clever(net, C.byref(p))
    n = net.contents
    ih = rc.view(n.ih_weights, n.ih_size)
    ho = rc.view(n.ho_weights, n.ho_size)
    sums = np.array([ih.astype(np.float64).sum(), np.abs(ih).astype(np.float64).sum(),
                     ho.astype(np.float64).sum(), np.abs(ho).astype(np.float64).sum()])
    assert np.array_equal(sums, Z[name])
    assert np.array_equal(ih[n.h_size:n.h_size + 256], Z[name + "_ih_head"])
    assert np.array_equal(ho[:256], Z[name + "_ho_head"])
    assert np.array_equal(np.array([n.rng.a, n.rng.b, n.rng.c, n.rng.d], dtype=np.uint64),
                          Z[name + "_rng"])
    AMD.rnn_delete_net(net)


@pytest.mark.parametrize("name,method", [("init_bottom_flat", rc.INIT_FLAT),
                                         ("init_bottom_fan_in", rc.INIT_FAN_IN)])
def test_bottom_layer_init_bit_exact_and_round_trips(name, method, tmp_path):
    """rnn_new_with_bottom_layer (recur-nn-init.c:158-219): the layer's layout, its
    weights drawn after the net's (recur-nn-init.c:566-572, 614-620), sharing by the
    clones (345-346) and the CDB records (recur-nn-io.c:109-120, 341-344)."""
    net = AMD.rnn_new_with_bottom_layer(42, 16, 39, 42, rc.FLAG_STANDARD, 5, None, 10, 1e-3, 0.9, 0.0,
                                        rc.RELU, 0)
    n = net.contents
    assert n.flags & rc.FLAG_BOTTOM_LAYER and n.input_size == 16
    bl = n.bottom_layer.contents
    assert (bl.input_size, bl.output_size, bl.i_size, bl.o_size) == (42, 16, 44, 16)
    assert bl.learn_rate_scale == 1.0 and not bl.aux
    addr = lambda p: C.addressof(p.contents)
    m = bl.i_size * bl.o_size  # the reference's carve-up of layer->mem (recur-nn-init.c:180-190)
    assert addr(bl.momentums) == addr(bl.mem) and addr(bl.inputs) == addr(bl.mem) + 4 * m
    assert addr(bl.weights) == addr(bl.inputs) + 4 * bl.i_size
    assert addr(bl.outputs) == addr(bl.weights) + 4 * m and addr(bl.delta) == addr(bl.outputs) + 4 * bl.o_size
    assert addr(bl.i_error) == addr(bl.delta) + 4 * m and addr(bl.o_error) == addr(bl.i_error) + 4 * bl.i_size
    p = rc.InitParams()
    AMD.rnn_init_default_weight_parameters(net, C.byref(p))
    p.method = method
    AMD.rnn_randomise_weights_clever(net, C.byref(p))
    w = rc.view(bl.weights, bl.i_size, bl.o_size).copy()
    assert np.array_equal(w, Z[name + "_b_w"])
    assert rc.view(n.ih_weights, n.ih_size).astype(np.float64).sum() == Z[name + "_ih_sum"][0]
    assert np.array_equal(np.array([n.rng.a, n.rng.b, n.rng.c, n.rng.d], dtype=np.uint64), Z[name + "_rng"])
    nets = AMD.rnn_new_training_set(net, 3)
    assert all(addr(nets[j].contents.bottom_layer) == addr(n.bottom_layer) for j in range(3))
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert AMD.rnn_save_net(net, b"bottom.net", 1) == 0
        again = AMD.rnn_load_net(b"bottom.net")
        assert again
        a = again.contents
        al = a.bottom_layer.contents
        assert (al.input_size, al.output_size, al.i_size, al.o_size) == (42, 16, 44, 16)
        assert np.array_equal(rc.view(al.weights, al.i_size, al.o_size), w)
        assert np.array_equal(rc.view(a.ih_weights, a.ih_size), rc.view(n.ih_weights, n.ih_size))
        keys = [k for k, _ in cdb_keys("bottom.net")]
        assert keys[-7:] == ["bottom_layer.input_size", "bottom_layer.output_size", "bottom_layer.i_size",
                             "bottom_layer.o_size", "bottom_layer.learn_rate_scale",
                             "bottom_layer.overlap", "bottom_layer.weights"]
        AMD.rnn_delete_net(again)
    finally:
        os.chdir(cwd)
    AMD.rnn_delete_training_set(nets, 3, 0)
    # a zero-sized bottom layer is no bottom layer (recur-nn-init.c:201-207)
    plain = AMD.rnn_new_with_bottom_layer(42, 0, 39, 42, rc.FLAG_STANDARD | rc.FLAG_BOTTOM_LAYER, 5, None,
                                          10, 1e-3, 0.9, 0.0, rc.RELU, 0)
    assert not plain.contents.bottom_layer and not (plain.contents.flags & rc.FLAG_BOTTOM_LAYER)
    assert plain.contents.input_size == 42
    AMD.rnn_delete_net(plain)


def test_training_set_clones_share_and_seed_like_reference():
    c = replay.golden_case("relu_weighted")  # its rng snapshot is after 30 steps without noise == after cloning
    import golden_cases as gc
    import scenarios as sc
    a = sc.ApiSet(AMD, **gc.case_kwargs(gc.TRAIN_CASES["relu_weighted"]))
    for j in range(a.S):
        n = a.nets[j].contents
        assert (n.rng.a, n.rng.b, n.rng.c, n.rng.d) == tuple(int(x) for x in c["rng"][j])
    n0, n1 = a.nets[0].contents, a.nets[1].contents
    addr = lambda p: C.addressof(p.contents)
    assert addr(n0.ih_weights) == addr(n1.ih_weights) and addr(n0.ho_weights) == addr(n1.ho_weights)
    b0, b1 = n0.bptt.contents, n1.bptt.contents
    for f in ("ih_momentum", "ho_momentum", "ih_delta", "ho_delta", "ih_delta_tmp"):
        assert addr(getattr(b0, f)) == addr(getattr(b1, f)), f
    assert addr(b0.history) != addr(b1.history) and addr(n0.hidden_layer) != addr(n1.hidden_layer)
    assert AMD.rnn_new_training_set(a.net, 0) is None or not AMD.rnn_new_training_set(a.net, 0)
    a.close()


def test_cdb_reads_reference_fixture_and_round_trips(tmp_path):
    fx = os.path.join(ROOT, "tests", "golden", "multi-text-6c34c563i73-h99-o3650.net")
    net = AMD.rnn_load_net(fx.encode())
    assert net
    n = net.contents
    # scalars of the committed reference net ("Checkpoint / resume")
    assert (n.input_size, n.hidden_size, n.output_size) == (73, 99, 3650)
    assert (n.i_size, n.h_size, n.o_size) == (176, 100, 3652)
    assert n.generation == 10659 and n.flags == 0x40053 and n.activation == rc.RESQRT
    b = n.bptt.contents
    assert b.depth == 50 and abs(b.learn_rate - 0.1) < 1e-7
    assert len(n.metadata) == 1050 and n.metadata.startswith(b"{")
    w = rc.view(n.ih_weights, n.ih_size).copy()
    assert np.isfinite(w).all() and np.abs(w).sum() > 100
    cwd = os.getcwd()
    os.chdir(tmp_path)  # rnn_save_net makes its temp file in the cwd (recur-nn-io.c:17-21)
    try:
        assert AMD.rnn_save_net(net, b"copy.net", 1) == 0
        assert not os.path.exists("copy.net~")  # the reference's backup never happens (recur-nn-io.c:130)
        again = AMD.rnn_load_net(b"copy.net")
        m = again.contents
        assert np.array_equal(rc.view(m.ih_weights, m.ih_size), w)
        assert np.array_equal(rc.view(m.ho_weights, m.ho_size), rc.view(n.ho_weights, n.ho_size))
        assert m.metadata == n.metadata and m.generation == n.generation
        assert (m.rng.a, m.rng.d) == (n.rng.a, n.rng.d)
        mb = m.bptt.contents
        assert (mb.index, mb.ho_scale, mb.min_error_factor) == (b.index, b.ho_scale, b.min_error_factor)
        # same record names in the same order as the reference writer
        keys = cdb_keys("copy.net")
        assert keys == cdb_keys(fx)
        assert AMD.rnn_load_net(b"no-such-file.net") is None or not AMD.rnn_load_net(b"no-such-file.net")
        assert AMD.rnn_save_net(None, b"x.net", 0) == -1
    finally:
        os.chdir(cwd)


def test_a_resaved_reference_net_is_the_reference_writers_file_byte_for_byte(tmp_path):
    """The write side of the container against the REFERENCE'S OWN WRITER: the committed .net was written by the
    reference (rnn_save_net over tinycdb's cdb_make, recur-nn-io.c:12-139); loaded and saved again by librecur_amd it is
    the same 1,535,309 bytes -- records, order, and both levels of hash tables as tinycdb lays them out, which is what
    rnn_load_net's cdb_seek walks (recur-nn-io.c:168-184)."""
    fx = os.path.join(ROOT, "tests", "golden", "multi-text-6c34c563i73-h99-o3650.net")
    net = AMD.rnn_load_net(fx.encode())
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert AMD.rnn_save_net(net, b"again.net", 0) == 0
        assert open("again.net", "rb").read() == open(fx, "rb").read()
    finally:
        os.chdir(cwd)


def test_written_nets_are_the_files_the_references_reader_read(tmp_path):
    """Files written by rnn_save_net, checked by the reference's own reader: tests/golden/make_cdb_reader_check.py (build
    container) opened them with /root/reference/scripts/pycdb.py and found EVERY record through the hash tables
    (pycdb.py:107-139; the fixture holds what it found).  Here the same nets are written again -- any box -- and must be
    those files to the byte, and this repo's sequential walk must see the records the reference's reader saw.  Where the
    reference tree is present the reader itself runs again on the fresh files."""
    import hashlib
    import json
    import cdb_cases
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "cdb_written_by_rnn_save_net.json")))
    assert set(want) == set(cdb_cases.CASES)
    for name in cdb_cases.CASES:
        path = cdb_cases.write_case(name, str(tmp_path))
        data = open(path, "rb").read()
        assert hashlib.sha256(data).hexdigest() == want[name]["sha256"] and len(data) == want[name]["bytes"], name
        assert cdb_keys(path) == [(r["key"], r["len"]) for r in want[name]["records"]]
        again = AMD.rnn_load_net(path.encode())  # and the library reads what it wrote
        assert again and again.contents.generation == (4242 if "metadata" in name else 0)
    if os.path.exists("/root/reference/scripts/pycdb.py"):
        import subprocess
        import sys
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_cdb_reader_check.py")],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and r.stdout.count("every key found by hash lookup") == 2, r.stdout + r.stderr[-2000:]
        assert json.load(open(os.path.join(ROOT, "tests", "golden", "cdb_written_by_rnn_save_net.json"))) == want


def cdb_keys(path):
    """Independent reader of the container (the layout scripts/pycdb.py of the
    reference documents): record keys in file order."""
    data = open(path, "rb").read()
    u32 = lambda o: int.from_bytes(data[o:o + 4], "little")
    end = min(u32(8 * t) for t in range(256))
    pos, keys = 2048, []
    while pos < end:
        kl, dl = u32(pos), u32(pos + 4)
        keys.append((data[pos + 8:pos + 8 + kl].decode(), dl))
        pos += 8 + kl + dl
    return keys


def test_compute_entry_points_fail_loudly_without_a_gpu():
    if AMD.rnn_amd_device_count() > 0:
        pytest.skip("a GPU is present")
    code = ("import sys; sys.path.insert(0, %r); import recur_ctypes as rc; a = rc.load_amd();"
            "n = a.rnn_new(4, 7, 3, rc.FLAG_STANDARD, 1, None, 4, 0.01, 0.9, 0.0, rc.RELU);"
            "a.rnn_opinion(n, None, 0.0)") % os.path.join(ROOT, "tests")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


def test_cold_host_functions_bit_exact_with_reference():
    """rnn_weight_noise, rnn_perforate_weights, rnn_zap_non_diagonals,
    rnn_clear_diagonal_only_section, rnn_scale_initial_weights, the FAN_IN / RUNS / ZERO
    initialisers, the optimiser ballast setters and rnn_log_net's lines
    (recur-nn.c:857-904, 1027-1145; recur-nn-init.c:359-380, 382-861) against
    tests/golden/ref_cold.npz, which tests/golden/make_golden_cold.py produced by running
    the same script (tests/cold_cases.py) on the compiled reference.  Weights AND the
    generator states must be bit-exact."""
    import cold_cases
    want = np.load(os.path.join(ROOT, "tests", "golden", "ref_cold.npz"))
    got = cold_cases.run(AMD)
    keys = [k for k in want.files if not k.startswith("shim.")]
    assert sorted(got) == sorted(keys) and len(keys) >= 60
    bad = [k for k in keys if not np.array_equal(got[k], want[k])]
    assert bad == []
    assert bytes(want["log_net.lines"]).decode().splitlines()[0] == "generation 77"


def test_delete_order_of_text_predict_does_not_touch_freed_engine():
    """text-predict.c:654-656 deletes the training set (prototype included) BEFORE its
    confabulation and validation clones; gstrnnca's teardown has the same shape.  The
    clones must survive their weight owner (a subprocess: a use-after-free may crash)."""
    code = ("import sys; sys.path.insert(0, %r); import recur_ctypes as rc; a = rc.load_amd();"
            "n = a.rnn_new(4, 7, 3, rc.FLAG_STANDARD, 1, None, 4, 0.01, 0.9, 0.0, rc.RELU);"
            "nets = a.rnn_new_training_set(n, 3);"
            "fl = n.contents.flags & ~(rc.FLAG_OWN_WEIGHTS | rc.FLAG_OWN_BPTT);"
            "c1 = a.rnn_clone(n, fl, rc.SUBSEED, None); c2 = a.rnn_clone(n, fl, rc.SUBSEED, None);"
            "a.rnn_delete_training_set(nets, 3, 0);"
            "a.rnn_delete_net(c1); a.rnn_delete_net(c2); print('ok')") % os.path.join(ROOT, "tests")
    for _ in range(3):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        assert r.returncode == 0 and "ok" in r.stdout, r.stderr


REF = "/root/reference"
needs_reference = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "recur-nn.h")),
                                     reason="build container only: needs /root/reference")

STRUCT_FIELDS = {
    "RecurNN": ["i_size", "h_size", "o_size", "input_size", "hidden_size", "output_size", "ih_size",
                "ho_size", "flags", "log", "mem", "input_layer", "hidden_layer", "output_layer",
                "ih_weights", "ho_weights", "real_inputs", "rng", "bptt", "bottom_layer", "metadata",
                "generation", "presynaptic_noise", "activation"],
    "RecurNNBPTT": ["depth", "index", "i_error", "h_error", "o_error", "ih_momentum", "ho_momentum",
                    "history", "ih_delta", "ho_delta", "ih_delta_tmp", "ih_aux", "ho_aux", "mem",
                    "learn_rate", "ih_scale", "ho_scale", "momentum", "momentum_weight",
                    "min_error_factor"],
    "RecurExtraLayer": ["mem", "weights", "momentums", "aux", "delta", "inputs", "outputs", "i_error",
                        "o_error", "learn_rate_scale", "input_size", "output_size", "i_size", "o_size",
                        "overlap"],
    "struct RecurInitialisationParameters": [
        "method", "submethod", "bias_uses_submethod", "inputs_use_submethod", "fan_in_sum",
        "fan_in_step", "fan_in_min", "fan_in_ratio", "flat_variance", "flat_shape", "flat_perforation",
        "run_input_probability", "run_input_magnitude", "run_gain", "run_len_mean", "run_len_stddev",
        "run_n", "run_loop", "run_crossing_paths", "run_inputs_miss", "run_input_at_start"],
    "RecurErrorRange": ["start", "len"],
}
ENUMS = ["RNN_NET_FLAG_OWN_BPTT", "RNN_NET_FLAG_OWN_WEIGHTS", "RNN_NET_FLAG_LOG_APPEND",
         "RNN_NET_FLAG_LOG_HIDDEN_SUM", "RNN_NET_FLAG_LOG_WEIGHT_SUM",
         "RNN_NET_FLAG_BPTT_ADAPTIVE_MIN_ERROR", "RNN_NET_FLAG_NO_MOMENTUMS", "RNN_NET_FLAG_NO_DELTAS",
         "RNN_NET_FLAG_BOTTOM_LAYER", "RNN_NET_FLAG_AUX_ARRAYS", "RNN_COND_USE_SCALE", "RNN_COND_USE_ZERO",
         "RNN_COND_USE_LAWN_MOWER", "RNN_COND_USE_TALL_POPPY", "RNN_COND_USE_RAND",
         "RNN_NET_FLAG_STANDARD", "RNN_MOMENTUM_WEIGHTED", "RNN_MOMENTUM_NESTEROV",
         "RNN_MOMENTUM_SIMPLIFIED_NESTEROV", "RNN_MOMENTUM_CLASSICAL", "RNN_ADAGRAD", "RNN_ADADELTA",
         "RNN_RPROP", "RNN_LAST_LEARNING_METHOD", "RNN_INIT_ZERO", "RNN_INIT_FLAT", "RNN_INIT_FAN_IN",
         "RNN_INIT_RUNS", "RNN_RELU", "RNN_RESQRT", "RNN_RECLIP20", "RNN_INIT_DIST_UNIFORM",
         "RNN_INIT_DIST_GAUSSIAN", "RNN_INIT_DIST_LOG_NORMAL", "RNN_INIT_DIST_SEMICIRCLE",
         "RNN_CONDITIONING_INTERVAL", "RNN_COND_USE_OFFSET"]


@needs_reference
def test_offsetof_table_equals_the_reference_header(tmp_path):
    """Every field of every public struct, every enum value: one C program compiled twice,
    once against the reference's recur-nn.h and once against include/recur-nn.h."""
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "recur-nn.h"', 'int main(void){']
    for st, fields in STRUCT_FIELDS.items():
        lines.append('printf("sizeof %s %%zu\\n", sizeof(%s));' % (st, st))
        for f in fields:
            lines.append('printf("%s.%s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s *)0)->%s));'
                         % (st, f, st, f, st, f))
    for e in ENUMS:
        lines.append('printf("%s %%ld\\n", (long)%s);' % (e, e))
    lines.append('printf("%g %g %g %g\\n", (double)MAX_TOP_ERROR_FACTOR, (double)BASE_MIN_ERROR_FACTOR, '
                 '(double)WEIGHT_SCALE, (double)RNN_MOMENTUM_WEIGHT);')
    lines.append("return 0;}")
    src = tmp_path / "offsets.c"
    src.write_text("\n".join(lines))
    outs = []
    for tag, inc in (("ref", ["-I", REF]), ("amd", ["-I", os.path.join(ROOT, "include")])):
        exe = tmp_path / ("offsets_" + tag)
        subprocess.run(["gcc", "-std=gnu11", "-D_GNU_SOURCE", "-fcommon", "-w"] + inc + [str(src), "-o", str(exe)],
                       check=True)
        outs.append(subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout)
    assert outs[0] == outs[1]
    assert outs[0].count("\n") > 120


@needs_reference
def test_reference_callers_compile_and_link_unchanged(tmp_path):
    """The drop-in claim, mechanically: the reference's own callers of this path --
    test/test_fb_backprop.c, text-predict.c with the reference's charmodel objects, and
    text-predict.c against the library's own rnn_char_* -- compile UNCHANGED with
    include/ in front of the reference's directory and link against librecur_amd.so
    (Makefile:118 lists the objects this replaces).  Nothing is copied; outputs go to tmp."""
    inc = os.path.join(ROOT, "include")
    lib = os.path.join(ROOT, "recur_amd", "lib")
    (tmp_path / "path.h").write_text('#define BASE_PATH "%s"\n#define TEST_DATA_DIR BASE_PATH "/test-images"\n'
                                     '#define TEST_VIDEO_DIR BASE_PATH "/test-video"\n'
                                     '#define DICKENS_SHUFFLED_TEXT BASE_PATH "/x.txt"\n'
                                     '#define DICKENS_TEXT BASE_PATH "/y.txt"\n'
                                     '#define EREWHON_TEXT BASE_PATH "/test-images/erewhon.txt"\n'
                                     '#define LYSISTRATA_TEXT BASE_PATH "/z.txt"\n' % REF)
    # ccan/opt wants a config.h (the reference generates it with ccan's configurator, Makefile:94-95)
    (tmp_path / "config.h").write_text("#define HAVE_TYPEOF 1\n#define HAVE_BUILTIN_TYPES_COMPATIBLE_P 1\n"
                                       "#define HAVE_STATEMENT_EXPR 1\n#define HAVE_ATTRIBUTE_UNUSED 1\n"
                                       "#define HAVE_ATTRIBUTE_NORETURN 1\n#define HAVE_ATTRIBUTE_COLD 1\n"
                                       "#define HAVE_ATTRIBUTE_PRINTF 1\n#define HAVE_ATTRIBUTE_CONST 1\n"
                                       "#define HAVE_ATTRIBUTE_PURE 1\n#define HAVE_ATTRIBUTE_USED 1\n"
                                       "#define HAVE_BUILTIN_CONSTANT_P 1\n#define HAVE_BUILTIN_EXPECT 1\n"
                                       "#define HAVE_SYS_TERMIOS_H 1\n#define HAVE_ASPRINTF 1\n")
    cflags = ["gcc", "-std=gnu11", "-O1", "-w", "-fcommon", "-D_GNU_SOURCE", "-DVECTOR", "-I", inc,
              "-I", str(tmp_path), "-I", REF, "-I", os.path.join(REF, "ccan", "opt")]
    link = ["-L", lib, "-lrecur_amd", "-Wl,-rpath," + lib, "-lm"]

    def cc(src, obj):
        subprocess.run(cflags + ["-c", src, "-o", str(tmp_path / obj)], check=True)
        return str(tmp_path / obj)

    # 1. the reference's backprop test program
    exe = tmp_path / "test_fb_backprop"
    subprocess.run(cflags + [os.path.join(REF, "test", "test_fb_backprop.c"), "-o", str(exe)] + link, check=True)
    # 2. text-predict.c + the reference's charmodel layer on the library's rnn_* symbols
    opt = [cc(os.path.join(REF, "ccan", "opt", f), "opt_" + f.replace(".c", ".o"))
           for f in ("opt.c", "parse.c", "helpers.c", "usage.c")]
    tp = cc(os.path.join(REF, "text-predict.c"), "text-predict.o")
    cm = [cc(os.path.join(REF, f), f.replace(".c", ".o")) for f in ("charmodel-predict.c", "charmodel-init.c")]
    subprocess.run(["gcc", tp] + cm + opt + ["-o", str(tmp_path / "text-predict-ref-charmodel")] + link, check=True)
    # 3. text-predict.c on the library's own rnn_char_* as well
    subprocess.run(["gcc", tp] + opt + ["-o", str(tmp_path / "text-predict-amd-charmodel")] + link, check=True)
    for name in ("test_fb_backprop", "text-predict-ref-charmodel", "text-predict-amd-charmodel"):
        r = subprocess.run(["nm", "-u", str(tmp_path / name)], capture_output=True, text=True, check=True)
        undefined = {ln.split()[-1].split("@")[0] for ln in r.stdout.splitlines()}
        assert sum(u.startswith("rnn_") for u in undefined) >= 5  # resolved by the shared library
    # the reference's --help runs (host only, no device call)
    r = subprocess.run([str(tmp_path / "text-predict-amd-charmodel"), "--help"], capture_output=True, text=True)
    assert "usage" in (r.stdout + r.stderr).lower()
