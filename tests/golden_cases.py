"""Scenario definitions shared by tests/golden/make_golden.py (which runs them on
the REAL reference, oracle/_ref/librecur_ref.so, in the build container) and by
the tests (which replay them on the oracle restatement and on librecur_amd.so
and compare with the stored reference results).

Each scenario is a dict of plain parameters; ``run_scenario`` drives any driver
object with the ApiSet / OracleSet interface of scenarios.py.
"""
import numpy as np

import recur_ctypes as rc

TEXT_LEN = 6000

# name -> parameters.  "steps" generations of the multi-tap text loop
# (charmodel-predict.c:288-311) from a freshly initialised net.
TRAIN_CASES = {
    "relu_weighted": dict(hidden=39, S=4, D=8, act=rc.RELU, method=rc.WEIGHTED, lr=1e-2, steps=30, seed=3),
    "relu_nesterov": dict(hidden=39, S=4, D=8, act=rc.RELU, method=rc.NESTEROV, lr=1e-2, steps=30, seed=3),
    "relu_adagrad": dict(hidden=39, S=4, D=8, act=rc.RELU, method=rc.ADAGRAD, lr=1e-2, steps=30, seed=3,
                         ballast=0.1),
    "resqrt_weighted": dict(hidden=39, S=4, D=8, act=rc.RESQRT, method=rc.WEIGHTED, lr=1e-2, steps=30, seed=4),
    "reclip_simplified": dict(hidden=39, S=4, D=8, act=rc.RECLIP20, method=rc.SIMPLIFIED_NESTEROV, lr=1e-2,
                              steps=30, seed=5),
    "relu_classical": dict(hidden=39, S=3, D=5, act=rc.RELU, method=rc.CLASSICAL, lr=3e-3, steps=20, seed=6),
    "relu_adadelta": dict(hidden=39, S=3, D=5, act=rc.RELU, method=rc.ADADELTA, lr=1e-3, steps=20, seed=7,
                          aux=True, momentum=0.95, ballast=0.0),
    "relu_rprop": dict(hidden=39, S=3, D=5, act=rc.RELU, method=rc.RPROP, lr=1e-3, steps=20, seed=8, aux=True,
                       aux_value=1e-4),
    # one generation only: a single forward/backward/update from a known state (G3)
    "single_step_h99": dict(hidden=99, S=3, D=10, act=rc.RESQRT, method=rc.WEIGHTED, lr=1e-3, steps=1, seed=9),
    "ragged_s7_h130": dict(hidden=130, S=7, D=6, act=rc.RELU, method=rc.WEIGHTED, lr=1e-3, steps=8, seed=10),
    # a learn rate high enough that the error-gain clamp (ih_scale < 1) and the
    # adaptive early exit both fire (recur-nn.c:387-413)
    "hot_clamps": dict(hidden=39, S=4, D=12, act=rc.RELU, method=rc.WEIGHTED, lr=0.08, steps=40, seed=11),
    # presynaptic noise: every stream draws from its own Jenkins generator
    # (recur-nn.c:120-121), so the final generator states are part of the parity
    "noisy": dict(hidden=39, S=4, D=6, act=rc.RELU, method=rc.WEIGHTED, lr=1e-2, steps=12, seed=15,
                  noise=0.05),
    # a bottom layer under the net (text-predict --bottom-layer): 42 symbols -> 16
    # rectified nodes -> 39 hidden.  bottom->o_error is never cleared on this path, so
    # the bottom deltas integrate every earlier stream and generation (recur-nn.c:377-382)
    "bottom_weighted": dict(hidden=39, S=4, D=8, act=rc.RELU, method=rc.WEIGHTED, lr=1e-2, steps=20,
                            seed=16, bottom=16, bottom_rate_scale=0.5),
    "bottom_adagrad_noisy": dict(hidden=39, S=3, D=6, act=rc.RESQRT, method=rc.ADAGRAD, lr=1e-2, steps=12,
                                 seed=17, bottom=12, ballast=0.1, noise=0.03),
    # ... and a bottom layer in the hot regime: a stream whose error gain is clipped shrinks the layer's
    # (shared, never cleared) error accumulator by ih_scale twice (recur-nn.c:391-399)
    "bottom_hot_clamps": dict(hidden=39, S=4, D=8, act=rc.RELU, method=rc.WEIGHTED, lr=0.05, steps=30,
                              seed=18, bottom=16, bottom_rate_scale=0.5),
    "depth1": dict(hidden=23, S=2, D=1, act=rc.RELU, method=rc.WEIGHTED, lr=1e-2, steps=6, seed=12),
}


def case_kwargs(c):
    flags = rc.FLAG_STANDARD | rc.FLAG_ADAPTIVE_MIN_ERROR
    if c.get("aux"):
        flags |= rc.FLAG_AUX_ARRAYS
    kw = dict(input_size=42, hidden_size=c["hidden"], output_size=42, S=c["S"], D=c["D"],
              activation=c["act"], learn_rate=c["lr"], seed=c["seed"], flags=flags,
              noise=c.get("noise", 0.0))
    if c.get("bottom"):
        kw.update(input_size=c["bottom"], bottom_inputs=42,
                  bottom_rate_scale=c.get("bottom_rate_scale", 1.0))
    return kw


from recur_amd.drivers import synthetic_text_np  # noqa: E402,F401


def prepare(driver, c, set_momentum_values, set_aux_values):
    """Optimiser ballast that some methods need before the first step."""
    if "ballast" in c and c["ballast"]:
        set_momentum_values(c["ballast"])
    if "aux_value" in c:
        set_aux_values(c["aux_value"])
        set_momentum_values(0.0)


def run_scenario(driver, c, text, trace=None):
    m = c.get("momentum", 0.9)
    for i in range(c["steps"]):
        driver.char_step(text, i, c["method"], m)
        if trace is not None:
            trace(i)
