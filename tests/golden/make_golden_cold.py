#!/usr/bin/env python3
"""Generates tests/golden/ref_cold.npz: the cold host-side functions of recur-nn.h
(tests/cold_cases.py) run on the REAL reference, oracle/_ref/librecur_ref.so
(compiled from /root/reference by oracle/Makefile).  Run in the build container:

    make -C oracle && python tests/golden/make_golden_cold.py

The file is pure data (the reference's outputs); no reference source travels.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import cold_cases  # noqa: E402
import recur_ctypes as rc  # noqa: E402


def main():
    ref = rc.load_ref()
    out = cold_cases.run(ref)
    # caller-side maths of the rnnca plugin that the oracle restates: fast_sigmoid
    # (badmaths.h:31-36) through oracle/ref_shim.c
    xs = np.concatenate([np.linspace(-40, 40, 161), [0.0, 0.2, -0.2, 1e-9, 88.0, -88.0]]).astype(np.float32)
    out["shim.fast_sigmoid_x"] = xs
    out["shim.fast_sigmoid_y"] = np.array([ref.ref_fast_sigmoid(float(x)) for x in xs], dtype=np.float32)
    path = os.path.join(HERE, "ref_cold.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%d arrays, %.1f KB" % (len(out), os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
