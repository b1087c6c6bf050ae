"""ctypes loader for oracle/geo_oracle.c (test infrastructure only)."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "geo_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B" if force else "-s"])
    return _SO


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        P = ctypes.c_void_p
        L.oracle_sssp.restype = ctypes.c_int
        L.oracle_sssp.argtypes = [ctypes.c_int32, P, P, P, P, P, P, ctypes.c_int, ctypes.c_int32, P, P, P]
        L.oracle_knn.restype = ctypes.c_int
        L.oracle_knn.argtypes = [P, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                 ctypes.c_int64, ctypes.c_int64, P, P]
        L.oracle_cc.restype = ctypes.c_int32
        L.oracle_cc.argtypes = [ctypes.c_int32, P, P, P, P, P]
        _lib = L
    return _lib
