"""ctypes loader for tests/cpu_emu/libpdwt_emu.so (CPU emulation of the HIP tile functions)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "cpu_emu")
f32p = C.POINTER(C.c_float)
_lib = None


def P(a):
    return a.ctypes.data_as(f32p)


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-s", "-C", EMU_DIR])
        _lib = C.CDLL(os.path.join(EMU_DIR, "libpdwt_emu.so"))
    return _lib


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)
