"""Development probe: stream-timesteps/s of the library's rnn_char_epoch on text-predict's default configuration
(BASELINE.json configs[0]: ONE net, 99 hidden units, BPTT depth 30, the single-net branch with rnn_bptt_calculate),
and of the same net through the multi-tap branch: gpu_epoch_rate.py [hidden] [depth] [steps]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import recur_ctypes as rc
import scenarios as sc

amd = rc.bind_char(rc.load_amd())
hidden = int(sys.argv[1]) if len(sys.argv) > 1 else 99
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 30
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6000
text = np.ascontiguousarray(sc.synthetic_text(steps + 600))
for multi_tap in (False, True):
    a = sc.ApiSet(amd, input_size=42, hidden_size=hidden, output_size=42, S=1, D=depth, learn_rate=1e-4, seed=3)
    model = rc.CharModel()
    model.net = a.net
    model.training_nets = a.nets
    model.n_training_nets = 1
    model.batch_size = 1
    model.momentum = 0.95
    model.momentum_soft_start = 0.0
    model.learning_style = rc.WEIGHTED
    model.report_interval = 1000
    model.save_net = False
    model.use_multi_tap_path = multi_tap
    amd.rnn_char_init_schedule(C.byref(model.schedule), 0, 0.0, 1.0, 0)
    v = rc.CharVentropy()
    amd.rnn_char_init_ventropy(C.byref(v), a.net, rc.u8ptr(text), 0, 1)
    amd.rnn_char_epoch(C.byref(model), None, C.byref(v), rc.u8ptr(text), len(text), 0, 500, 0.0, 0, -1, 2, 0, 0)  # warm-up
    amd.rnn_amd_synchronize()
    t0 = time.perf_counter()
    amd.rnn_char_epoch(C.byref(model), None, C.byref(v), rc.u8ptr(text), len(text), 500, 500 + steps, 0.0, 0, -1, 2, 0, 0)
    amd.rnn_amd_synchronize()
    dt = time.perf_counter() - t0
    print("rnn_char_epoch, one net, hidden %d, depth %d, %s branch: %.0f stream-timesteps/s (%.1f us per step)" % (
        hidden, depth, "multi-tap" if multi_tap else "single-net (rnn_bptt_calculate)", steps / dt, 1e6 * dt / steps))
    amd.rnn_char_delete_ventropy(C.byref(v))
    a.close()
