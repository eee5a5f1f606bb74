"""recur_amd -- Python-side loader for librecur_amd.so (the product is the C
library; this package only locates it and checks its exported surface)."""
import ctypes
import os

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "librecur_amd.so")


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("librecur_amd.so is not built; run __graft_entry__.build()")
    return ctypes.CDLL(LIB_PATH)


_lib = load()
