"""Pins oracle/liboracle.so (the repo's CPU restatement) to vectors produced by
the real reference (tests/golden/ref_vectors.npz, see golden/make_golden.py).
Integer results are bit-exact; fp32 results agree to 2e-6 relative (2-norm):
same operation order, a strict-IEEE build on both sides."""
import ctypes as C

import numpy as np
import pytest

import golden_cases as gc
import recur_ctypes as rc
import replay

ORC = rc.load_oracle()
Z = replay.golden()
RTOL = 2e-6


@pytest.mark.parametrize("seed", [1, 2, 11])
def test_rand64_bit_exact(seed):
    ctx = rc.OrcRng()
    ORC.orc_init_rand64(C.byref(ctx), seed)
    got = np.array([ORC.orc_rand64(C.byref(ctx)) for _ in range(64)], dtype=np.uint64)
    assert np.array_equal(got, Z["rand64_seed%d" % seed])


def test_prng_derived_streams_bit_exact():
    ctx = rc.OrcRng()
    ORC.orc_init_rand64(C.byref(ctx), 1)
    g = np.array([ORC.orc_cheap_gaussian_noise(C.byref(ctx)) for _ in range(32)], dtype=np.float32)
    d = np.array([ORC.orc_rand_double(C.byref(ctx)) for _ in range(16)])
    i = np.array([ORC.orc_rand_small_int(C.byref(ctx), 1000) for _ in range(16)], dtype=np.int32)
    assert np.array_equal(g, Z["gaussian_seed1"])
    assert np.array_equal(d, Z["rand_double_seed1"])
    assert np.array_equal(i, Z["rand_small_int_seed1"])


def test_synthetic_text_matches_numpy_restatement():
    import scenarios as sc
    assert np.array_equal(sc.synthetic_text_oracle(gc.TEXT_LEN), gc.synthetic_text_np())


def test_fast_expf_softmax_softclip():
    y = np.array([ORC.orc_fast_expf(float(x)) for x in Z["expf_x"]], dtype=np.float32)
    assert np.array_equal(y, Z["expf_y"])
    sm_in = Z["softmax_in"]
    for i in range(sm_in.shape[0]):
        out = np.zeros(44, np.float32)
        err = np.zeros(44, np.float32)
        ORC.orc_softmax(rc.fptr(out), rc.fptr(np.ascontiguousarray(sm_in[i])), 42)
        best = ORC.orc_softmax_best_guess(rc.fptr(err), rc.fptr(np.ascontiguousarray(sm_in[i])), 42)
        assert np.array_equal(out, Z["softmax_out"][i])
        assert np.array_equal(err, Z["softmax_err"][i])
        assert best == Z["softmax_best"][i]
    got = np.array([ORC.orc_soft_clip(float(s), 2.0) for s in Z["soft_clip_in"]], dtype=np.float32)
    assert np.array_equal(got, Z["soft_clip_out"])
    ss = np.array([ORC.orc_momentum_soft_start(g, 0.95, x) for g in (0, 10, 1000, 1e6)
                   for x in (0, 1, 2000)], dtype=np.float32)
    assert np.array_equal(ss, Z["soft_start"])


@pytest.mark.parametrize("name,hidden,shape,perf,seed", [
    ("init_semicircle_h99", 99, rc.DIST_SEMICIRCLE, 0.0, 1),
    ("init_uniform_perf_h99", 99, rc.DIST_UNIFORM, 0.7, 1),
    ("init_gaussian_h1024", 1024, rc.DIST_GAUSSIAN, 0.0, 2),
    ("init_lognormal_h64", 64, rc.DIST_LOG_NORMAL, 0.2, 3),
])
def test_flat_init_bit_exact(name, hidden, shape, perf, seed):
    z = ORC.orc_set_new(42, hidden, 42, 1, 10, rc.RELU, rc.FLAG_STANDARD, 1e-3, seed)
    H = z.contents.H
    ORC.orc_set_init_flat(z, np.float32(2.0) / np.float32(H), shape, perf)
    ih = rc.view(z.contents.ih_w, z.contents.I * H)
    ho = rc.view(z.contents.ho_w, H * z.contents.O)
    sums = np.array([ih.astype(np.float64).sum(), np.abs(ih).astype(np.float64).sum(),
                     ho.astype(np.float64).sum(), np.abs(ho).astype(np.float64).sum()])
    assert np.array_equal(sums, Z[name])
    assert np.array_equal(ih[H:H + 256], Z[name + "_ih_head"])
    assert np.array_equal(ho[:256], Z[name + "_ho_head"])
    r = z.contents.rng[0]
    assert np.array_equal(np.array([r.a, r.b, r.c, r.d], dtype=np.uint64), Z[name + "_rng"])
    ORC.orc_set_free(z)


@pytest.mark.parametrize("name", sorted(gc.TRAIN_CASES))
def test_training_scenarios(name):
    got = replay.train_oracle(name)
    want = replay.golden_case(name)
    replay.check(got, want, RTOL, exact=("index", "generation", "rng"))


def test_hot_case_exercises_the_clamps():
    """the 'hot_clamps' fixture really contains ih_scale < 1 steps, so the
    error-gain branch (recur-nn.c:393-402) is covered by the parity tests"""
    tr = Z["hot_clamps.ih_scale_trace"]
    assert tr.min() < 0.5 and (tr == 1.0).any()


@pytest.mark.parametrize("bit,flag", replay.COND_BITS)
def test_conditioning(bit, flag):
    got = replay.cond_oracle(bit, flag)
    for k, v in got.items():
        assert np.array_equal(v, Z["cond%d.%s" % (bit, k)]), k


@pytest.mark.parametrize("name,batch", [("fused_b1", 1), ("fused_b4", 4)])
def test_fused_single_net_path(name, batch):
    replay.check(replay.fused_oracle(batch), replay.golden_case(name), RTOL)


def test_sparse_error_ranges():
    replay.check(replay.sparse_oracle(), replay.golden_case("sparse"), RTOL)


def test_fast_sigmoid_matches_reference():
    """badmaths.h:31-36 through oracle/ref_shim.c (tests/golden/ref_cold.npz): what
    orc_sigmoid_mse_error, the restatement of rnnca's loss (gstrnnca.c:701-714), is made of."""
    import os
    z = np.load(os.path.join(rc.ROOT, "tests", "golden", "ref_cold.npz"))
    orc = rc.load_oracle()
    got = np.array([orc.orc_fast_sigmoid(float(x)) for x in z["shim.fast_sigmoid_x"]], dtype=np.float32)
    assert np.array_equal(got, z["shim.fast_sigmoid_y"])
