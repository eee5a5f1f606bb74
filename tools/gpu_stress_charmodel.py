"""Development probe: rnn_char_cross_entropy / rnn_char_prime (charmodel-predict.c:62-80, 407-431) in random sequences,
with short and degenerate lengths, on a training net and on a forward-only clone of random shape, against the oracle's
restatement; the hidden state compared after every call: gpu_stress_charmodel.py <seed> <trials>"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import recur_ctypes as rc
import scenarios as sc

amd = rc.bind_char(rc.load_amd())
rs = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10):
    A = int(rs.integers(2, 70))
    hidden = int(rs.choice([9, 17, 40, 99, 128, 256, 300]))
    kw = dict(input_size=A, hidden_size=hidden, output_size=A, S=1, D=int(rs.integers(1, 6)), learn_rate=1e-3,
              seed=int(rs.integers(1, 1000)), activation=int(rs.choice([rc.RELU, rc.RESQRT, rc.RECLIP20])))
    use_clone = bool(rs.integers(0, 2))
    print("next:", kw, "clone" if use_clone else "training net", flush=True)
    g = sc.ApiSet(amd, **kw)
    o = sc.OracleSet(**kw)
    net = g.net
    if use_clone:
        fl = g.net.contents.flags & ~(rc.FLAG_OWN_BPTT | rc.FLAG_OWN_WEIGHTS)
        net = amd.rnn_clone(g.net, fl, 5, None)
    res, log = "ok", []
    try:
        for step in range(int(rs.integers(2, 9))):
            if rs.random() < 0.5:
                n = int(rs.choice([2, 3, 4, int(rs.integers(2, 120))]))
                ign = int(rs.integers(0, max(1, min(n - 1, 9))))
                text = np.ascontiguousarray(rs.integers(0, A, n).astype(np.uint8))
                got = amd.rnn_char_cross_entropy(net, None, rc.u8ptr(text), n, ign, None, 0)
                want = o.orc.orc_cross_entropy(o.z, 0, rc.u8ptr(text), n, ign)
                log.append("xent(%d,%d)" % (n, ign))
                assert abs(got - want) <= 2e-4 * abs(want) + 1e-9, ("cross entropy", got, want)
            else:
                n = int(rs.choice([1, 2, 3, int(rs.integers(1, 40))]))
                text = np.ascontiguousarray(rs.integers(0, A, n).astype(np.uint8))
                last = amd.rnn_char_prime(net, None, rc.u8ptr(text), n)
                for i in range(n - 1):
                    o.orc.orc_one_hot_opinion(o.z, 0, int(text[i]), 0.0)
                log.append("prime(%d)" % n)
                assert last == int(text[-1]), ("prime's return", last, int(text[-1]))
            amd.rnn_amd_sync_host(net, rc.RNN_AMD_STREAM)
            hid = rc.view(net.contents.hidden_layer, g.H)
            want_h = o.arrays()["hidden"][0]
            d = np.abs(hid - want_h).max() / max(np.abs(want_h).max(), 1e-30)
            assert d <= 2e-4, ("hidden state after " + log[-1], d)
    except AssertionError as e:
        res = "MISMATCH after %s: %s" % (log, str(e)[:200])
        bad += 1
    print("   %s" % res, flush=True)
    if use_clone:
        amd.rnn_delete_net(net)
    g.close()
    o.close()
print("bad:", bad)
