"""Python handle on the CPU oracle (oracle/liboracle_isls.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The marshaling class is the product's own (`isls._capi.Kernels`), bound here to the `oracle_*`
symbols with host (numpy) pointers and no stream, so HIP path and oracle receive identical arguments.
"""
import ctypes
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(os.path.dirname(_HERE), "ilqr-admm_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from isls._capi import Kernels  # noqa: E402

LIB_PATH = os.path.join(_HERE, "liboracle_isls.so")


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return LIB_PATH


def load():
    build()
    lib = ctypes.CDLL(LIB_PATH)
    return Kernels(lib, prefix="oracle_", with_stream=False), lib


def set_threads(lib, n):
    lib.oracle_set_threads.restype = ctypes.c_int
    return lib.oracle_set_threads(ctypes.c_int(n))
