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
    out = cold_cases.run(rc.load_ref())
    path = os.path.join(HERE, "ref_cold.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%d arrays, %.1f KB" % (len(out), os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
