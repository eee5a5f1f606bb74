"""The prediction layer of the character models on the device
(recur_amd/csrc/char_epoch.c, char_sampling.c, char_multitext.c: rnn_char_epoch, rnn_char_cross_entropy,
rnn_char_prime, rnn_char_confabulate, the validation-entropy object), i.e. SURVEY.md
section 8(f) rank 1: the callers either side of the hot path, driven exactly the way
text-predict.c:529-640 drives them."""
import ctypes as C
import os

import numpy as np
import pytest

import erewhon_case as ec
import recur_ctypes as rc
import replay
import scenarios as sc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    lib = rc.bind_char(rc.load_amd())
    assert lib.rnn_amd_device_count() >= 1, "no HIP device: the product has no CPU fallback"
    return lib


def make_text_net(lib, log_file=None, **over):
    kw = dict(ec.KW)
    kw.update(over)
    a = sc.ApiSet(lib, **kw)
    if log_file:
        lib.rnn_set_log_file(a.net, log_file.encode(), 0)
    return a


def forward_clone(lib, net):
    # text-predict.c:538-541: borrows the weights, no bptt
    return lib.rnn_clone(net, net.contents.flags & ~(rc.FLAG_OWN_BPTT | rc.FLAG_OWN_WEIGHTS), rc.SUBSEED, None)


def oracle_with_weights_of(a, S=1, D=1):
    a.sync()
    n = a.net.contents
    o = sc.OracleSet(input_size=a.input_size, hidden_size=a.hidden_size, output_size=a.output_size, S=S, D=D,
                     learn_rate=1e-3, seed=1)
    o.arrays()["ih_w"][:] = rc.view(n.ih_weights, a.I, a.H)
    o.arrays()["ho_w"][:] = rc.view(n.ho_weights, a.H, a.O)
    return o


def test_epoch_driver_reproduces_the_reference_curve(amd, tmp_path):
    """rnn_char_epoch with the multi-tap branch on the device against the golden curve the
    real reference produced (tests/golden/make_golden.py, 'erewhon.*'): training entropy per
    250-generation report from the net's log file, validation entropy from the
    RnnCharVentropy object, stop after 1500 generations."""
    lib = amd
    text = ec.encoded_text()
    train = np.ascontiguousarray(text[:-ec.VALIDATE])
    valid = np.ascontiguousarray(text[-ec.VALIDATE:])
    log = str(tmp_path / "epoch.log")
    a = make_text_net(lib, log_file=log)
    vnet = forward_clone(lib, a.net)
    model = rc.CharModel()
    model.net = a.net
    model.training_nets = a.nets
    model.n_training_nets = a.S
    model.batch_size = 1
    model.momentum = 0.95
    model.momentum_soft_start = 0.0
    model.learning_style = rc.WEIGHTED
    model.report_interval = ec.WINDOW
    model.save_net = False
    model.use_multi_tap_path = True
    lib.rnn_char_init_schedule(C.byref(model.schedule), 0, 0.0, 1.0, 0)
    v = rc.CharVentropy()
    lib.rnn_char_init_ventropy(C.byref(v), vnet, rc.u8ptr(valid), len(valid), 1)
    done = lib.rnn_char_epoch(C.byref(model), None, C.byref(v), rc.u8ptr(train), len(train), 0,
                              ec.GENERATIONS, 0.0, 0, -1, 2, 0, 0)
    assert done == 1 and a.net.contents.generation == ec.GENERATIONS
    lib.rnn_set_log_file(a.net, None, 0)
    lines = [ln.split() for ln in open(log)]
    t_entropy = np.array([float(x[1]) for x in lines if x[0] == "t_entropy"])
    v_entropy = np.array([float(x[1]) for x in lines if x[0] == "v_entropy"])
    per_second = [float(x[1]) for x in lines if x[0] == "per_second"]
    z = replay.golden()
    assert len(t_entropy) == ec.GENERATIONS // ec.WINDOW
    assert np.allclose(t_entropy, z["erewhon.t_entropy"], rtol=1e-2), (t_entropy, z["erewhon.t_entropy"])
    # the validation net keeps its hidden state between reports (as the reference's does), the
    # golden value starts from a fresh clone: the 4000-symbol mean differs by far less than 2 %
    assert abs(v_entropy[-1] - z["erewhon.v_entropy"][0]) < 0.02 * z["erewhon.v_entropy"][0]
    assert abs(v.entropy - v_entropy[-1]) < 1e-4 and all(p > 0 for p in per_second)
    # every generation of every stream logged by the core too (recur-nn.c:766-771)
    assert sum(1 for x in lines if x[0] == "generation") >= ec.GENERATIONS
    lib.rnn_char_delete_ventropy(C.byref(v))
    lib.rnn_delete_net(vnet)
    a.close()


def test_cross_entropy_and_prime_match_oracle(amd):
    """get_cross_entropy / rnn_char_prime (charmodel-predict.c:62-80, 407-431) on the device, for a
    forward-only clone and for a training net, against the oracle's restatement."""
    lib = amd
    text = ec.encoded_text()
    a = make_text_net(lib)
    train = np.ascontiguousarray(text[:20000])
    a._sbg = rc.load_oracle().orc_softmax_best_guess
    for i in range(60):                       # a few generations so that the weights are not the initial ones
        a.char_step(train, i, rc.WEIGHTED, 0.9)
    seg = np.ascontiguousarray(text[30000:30600])
    prefix = np.ascontiguousarray(text[29000:29100])
    o = oracle_with_weights_of(a)
    clone = forward_clone(lib, a.net)
    got = lib.rnn_char_cross_entropy(clone, None, rc.u8ptr(seg), len(seg), 5, None, 0)
    want = o.orc.orc_cross_entropy(o.z, 0, rc.u8ptr(seg), len(seg), 5)
    assert abs(got - want) < 1e-4 * want, (got, want)
    # a second call continues from the state the first one left (no reset), with a primer first
    assert lib.rnn_char_prime(clone, None, rc.u8ptr(prefix), len(prefix)) == int(prefix[-1])
    for i in range(len(prefix) - 1):
        o.orc.orc_one_hot_opinion(o.z, 0, int(prefix[i]), 0.0)
    got2 = lib.rnn_char_cross_entropy(clone, None, rc.u8ptr(seg), len(seg), 0, None, 0)
    want2 = o.orc.orc_cross_entropy(o.z, 0, rc.u8ptr(seg), len(seg), 0)
    assert abs(got2 - want2) < 1e-4 * want2 and abs(got2 - got) > 1e-6
    lib.rnn_amd_sync_host(clone, rc.RNN_AMD_STREAM)
    hid = rc.view(clone.contents.hidden_layer, a.H)
    assert rc.rel_err(hid, o.arrays()["hidden"][0]) < 1e-4
    # degenerate lengths: nothing to score
    assert lib.rnn_amd_run_text(clone, rc.u8ptr(seg), 1, 0) == 0.0
    assert lib.rnn_char_prime(clone, None, None, 0) == 0
    lib.rnn_delete_net(clone)
    o.close()
    a.close()


def test_confabulation_follows_the_nets_generator(amd):
    """rnn_char_confabulate (charmodel-predict.c:137-181): sampling from the biased softmax
    (pinned on the reference's header in test_oracle_golden) with draws from the confabulating
    net's own generator; restated here on the oracle."""
    lib = amd
    a = make_text_net(lib)
    conf = forward_clone(lib, a.net)
    alphabet = rc.default_text_alphabet(lib)
    o = oracle_with_weights_of(a)
    orc = o.orc
    g = rc.OrcRng()
    n = conf.contents
    g.a, g.b, g.c, g.d = n.rng.a, n.rng.b, n.rng.c, n.rng.d
    for bias, start, stop in ((1.0, -1, -1), (0.0, -1, -1), (200.0, -1, -1)):
        prev = C.c_int(3)
        buf = C.create_string_buffer(400)
        wrote = lib.rnn_char_confabulate(conf, buf, 60, 400, alphabet, bias, C.byref(prev), start, stop)
        want, hot = [], 3
        tmp, p = np.zeros(42, np.float32), np.zeros(42, np.float32)
        for _ in range(60):
            ans = orc.orc_one_hot_opinion(o.z, 0, hot, 0.0)
            ans = np.ctypeslib.as_array(ans, shape=(42,)).copy()
            if bias >= 100:
                hot = int(42 - 1 - np.argmax(ans[::-1]))     # the last of equal maxima (">=")
            else:
                orc.orc_softmax(rc.fptr(tmp), rc.fptr(ans), 42)
                if bias:
                    tmp = (tmp * np.float32(bias) + ans).astype(np.float32)
                    orc.orc_softmax(rc.fptr(p), rc.fptr(tmp), 42)
                else:
                    p[:] = tmp
                hot = -1
                while hot < 0:
                    r = np.float32(orc.orc_rand_double(C.byref(g)))
                    acc = np.float32(0)
                    for i in range(42):
                        acc = np.float32(acc + p[i])
                        if r < acc:
                            hot = i
                            break
            want.append(rc.DEFAULT_CHARSET[hot])
        assert buf.value == bytes(want) and wrote == 60 and prev.value == hot
    assert (n.rng.a, n.rng.d) == (g.a, g.d)
    # too little room: nothing is written
    small = C.create_string_buffer(1)
    assert lib.rnn_char_confabulate(conf, small, 10, 1, alphabet, 1.0, C.byref(C.c_int(0)), -1, -1) == 0
    lib.rnn_char_free_alphabet(alphabet)
    lib.rnn_delete_net(conf)
    o.close()
    a.close()


def test_epoch_driver_single_net_branch_matches_oracle(amd, tmp_path):
    """BASELINE.json configs[0] shape: text-predict's default path, one net, the fused
    rnn_bptt_calculate branch of rnn_char_epoch (charmodel-predict.c:312-321) with temporal
    batching, against the oracle's restatement of the same loop."""
    lib = amd
    text = np.ascontiguousarray(ec.encoded_text()[:4000])
    for batch in (1, 5):
        a = make_text_net(lib, hidden_size=39, S=1, D=8, learn_rate=3e-3, seed=71)
        o = sc.OracleSet(input_size=42, hidden_size=39, output_size=42, S=1, D=8, learn_rate=3e-3, seed=71)
        model = rc.CharModel()
        model.net = a.net
        model.training_nets = a.nets
        model.n_training_nets = 1
        model.batch_size = batch
        model.momentum = 0.9
        model.momentum_soft_start = 0.0
        model.learning_style = rc.WEIGHTED
        model.report_interval = 100
        model.use_multi_tap_path = False
        lib.rnn_char_init_schedule(C.byref(model.schedule), 0, 0.0, 1.0, 0)
        v = rc.CharVentropy()
        lib.rnn_char_init_ventropy(C.byref(v), a.net, rc.u8ptr(text), 0, 1)      # no validation text
        done = lib.rnn_char_epoch(C.byref(model), None, C.byref(v), rc.u8ptr(text), len(text), 0, 150, 0.0, 0, -1,
                                  2, 0, 0)
        assert done == 1 and a.net.contents.generation == 150
        for i in range(150):
            o.orc.orc_advance(o.z, 0)
            o.orc.orc_net_error_bptt(o.z, 0, int(text[i]), int(text[i + 1]), C.byref(C.c_int(0)))
            o.orc.orc_bptt_calculate(o.z, 0, batch, 0.9)
        sg, so = a.snapshot(), o.snapshot()
        bad = sc.compare(sg, so, 1e-3, keys=["ih_w", "ho_w", "ih_m", "ho_m", "hidden", "output"],
                         exact=["index", "generation"])
        assert not bad, (batch, bad)
        lib.rnn_char_delete_ventropy(C.byref(v))
        a.close()
        o.close()


@pytest.mark.parametrize("hidden,D,batch,noise", [(99, 30, 1, 0.0), (39, 8, 4, 0.02), (256, 6, 1, 0.0), (130, 5, 3, 0.0)])
def test_fused_single_net_step_on_the_device_matches_oracle(amd, hidden, D, batch, noise):
    """rnn_amd_set_char_step_fused: the single-net branch of rnn_char_epoch (charmodel-predict.c:312-321: advance,
    one_hot_opinion, net_error_bptt's loss, rnn_bptt_calculate with temporal batching) without a host round trip
    per symbol -- what text-predict's default configuration (BASELINE.json configs[0]: one net, 99 hidden units,
    depth 30) runs through the library's rnn_char_epoch.  Step by step against the oracle, loss statistics included."""
    lib = amd
    kw = dict(input_size=42, hidden_size=hidden, output_size=42, S=1, D=D, learn_rate=3e-3, seed=33, noise=noise)
    a = sc.AmdBatchedSet(lib, **kw)
    o = sc.OracleSet(**kw)
    text = sc.synthetic_text(3000)
    a.load_text(text)
    a.stats(clear=True)
    err = ent = 0.0
    correct = 0
    steps = D + 9
    for i in range(steps):
        m = 0.9 if i % 2 else 0.8                      # the epoch's momentum soft start changes it per step
        a.net.contents.bptt.contents.momentum = m
        lib.rnn_amd_set_char_step_fused(a.handle, i, batch)
        c = C.c_int(0)
        o.orc.orc_advance(o.z, 0)
        e = o.orc.orc_net_error_bptt(o.z, 0, int(text[i]), int(text[i + 1]), C.byref(c))
        o.orc.orc_bptt_calculate(o.z, 0, batch, m)
        err += e
        ent += np.log2(max(1.0 - e, 1e-30))
        correct += c.value
    sg, so = a.snapshot(), o.snapshot()
    assert np.array_equal(sg["hidden"] != 0, so["hidden"] != 0)
    replay.check(sg, so, 1e-4, keys=["ih_w", "ho_w", "ih_m", "ho_m", "ih_delta", "hidden", "output", "hist", "o_error",
                                     "min_error_factor", "ih_scale"], exact=("index", "generation", "rng"))
    st = a.stats()
    assert st.count == steps and st.correct == correct
    assert abs(st.error - err) < 1e-4 * abs(err) and abs(st.entropy - ent) < 1e-3 * abs(ent)
    a.close()
    o.close()
