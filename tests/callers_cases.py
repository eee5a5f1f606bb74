"""Inputs of the caller-loss fixtures (tests/golden/make_golden_callers.py writes the reference's
results for them into tests/golden/ref_callers.npz; tests/test_oracle_callers_golden.py replays them
on the oracle).  Pure data and seeds: nothing here computes a loss."""
import numpy as np

import recur_ctypes as rc

MULTI_CASES = {
    # alphabet A, `classes` heads; every pass trains one class's stretch of the text
    "multi_plain": dict(A=11, classes=4, hidden=23, D=5, lr=5e-3, seed=31, activation=rc.RELU, noise=0.0,
                        leakage=0.0, method=rc.WEIGHTED, batch=1, passes=3, steps=40),
    "multi_leaky_batched": dict(A=7, classes=6, hidden=30, D=6, lr=0.05, seed=32, activation=rc.RESQRT, noise=0.0,
                                leakage=0.35, method=rc.ADAGRAD, batch=5, passes=4, steps=37, ballast=200.0),
    "multi_noisy": dict(A=13, classes=3, hidden=19, D=4, lr=4e-3, seed=33, activation=rc.RELU, noise=0.05,
                        leakage=0.6, method=rc.NESTEROV, batch=7, passes=3, steps=33),
}


def multi_kwargs(c):
    """constructor arguments of the net (the same for the reference's driver and the oracle's)"""
    flags = rc.FLAG_STANDARD | rc.FLAG_ADAPTIVE_MIN_ERROR
    return dict(input_size=c["A"], hidden_size=c["hidden"], output_size=c["A"] * c["classes"], S=1, D=c["D"],
                learn_rate=c["lr"], seed=c["seed"], activation=c["activation"], noise=c["noise"], flags=flags)


def multi_text(c):
    rng = np.random.default_rng(c["seed"] * 7 + 1)
    return rng.integers(0, c["A"], size=c["passes"] * c["steps"], dtype=np.uint8)


def multi_passes(c, n):
    """(target class, first symbol, one past the last) of each training pass"""
    return [((p * 5 + 1) % c["classes"], p * c["steps"], (p + 1) * c["steps"]) for p in range(c["passes"])]


GROUP_CASES = {
    "groups_two_classes": dict(sizes=[2], streams=12, seed=41, weighted=False, skip=0.25),
    "groups_mixed": dict(sizes=[3, 2, 5], streams=16, seed=42, weighted=True, skip=0.3),
}


def group_inputs(c):
    rng = np.random.default_rng(c["seed"])
    O = int(sum(c["sizes"]))
    answers = (rng.standard_normal((c["streams"], O)) * rng.choice([0.5, 3.0, 30.0], size=(c["streams"], 1))).astype(
        np.float32)
    targets = np.stack([rng.integers(0, n, size=c["streams"]) for n in c["sizes"]], axis=1).astype(np.int32)
    targets[rng.random(targets.shape) < c["skip"]] = -1  # unknown target / ignored window: the group is not trained
    targets[0, :] = -1  # a stream with nothing to train
    weight = (0.25 + rng.random(O)).astype(np.float32) if c["weighted"] else None
    return answers, targets, weight


def sigmoid_inputs():
    rng = np.random.default_rng(51)
    answers = (rng.standard_normal((10, 3)) * np.array([0.3, 2.0, 12.0])).astype(np.float32)
    answers[0] = (0.0, -70.0, 70.0)
    targets = rng.random((10, 3)).astype(np.float32)
    return answers, targets
