"""ctypes front end of the CPU oracle of the geometric verification (oracle/geometry_oracle.cc).

Test infrastructure only: imported by tests/, never by the sfm_amd package.  Parity with the
reference is unpinned and can only be statistical (the reference seeds its sampler from
std::random_device); the header of geometry_oracle.cc says what is followed literally.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libgeometry_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        l = ctypes.CDLL(_LIB)
        vp, i64, u64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64
        l.geo_sample.argtypes = [i64, u64, u64, u64, vp]
        l.geo_sample.restype = None
        l.geo_fit.argtypes = [vp, vp, vp, vp]
        l.geo_fit.restype = None
        l.geo_error.argtypes = [vp, vp, vp]
        l.geo_error.restype = ctypes.c_double
        l.geo_rounds_needed.argtypes = [i64, i64]
        l.geo_rounds_needed.restype = i64
        l.geo_compute_fundamental.argtypes = [vp, vp, i64, u64, u64, ctypes.c_int, vp, vp, vp, vp]
        l.geo_compute_fundamental.restype = ctypes.c_int
        _lib = l
    return _lib


def _xy(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    assert a.ndim == 2 and a.shape[1] == 2, a.shape
    return a


def sample(n, seed, pair, hypothesis):
    out = np.empty(8, np.int32)
    lib().geo_sample(int(n), int(seed), int(pair), int(hypothesis), out.ctypes.data)
    return out


def fit(lhs_xy, rhs_xy, idx8):
    l, r = _xy(lhs_xy), _xy(rhs_xy)
    idx = np.ascontiguousarray(idx8, dtype=np.int32)
    F = np.empty(9, np.float64)
    lib().geo_fit(l.ctypes.data, r.ctypes.data, idx.ctypes.data, F.ctypes.data)
    return F.reshape(3, 3)


def error(lhs_xy, rhs_xy, F):
    l = np.ascontiguousarray(lhs_xy, dtype=np.float64)
    r = np.ascontiguousarray(rhs_xy, dtype=np.float64)
    f = np.ascontiguousarray(F, dtype=np.float64)
    return lib().geo_error(l.ctypes.data, r.ctypes.data, f.ctypes.data)


def rounds_needed(inliers, n):
    return int(lib().geo_rounds_needed(int(inliers), int(n)))


def compute_fundamental(lhs_xy, rhs_xy, seed=0, pair=0, sampler="counter"):
    """(ok, F [3, 3], inlier flags uint8 [n], rounds) as ComputeFundamentalMatrix decides them."""
    l, r = _xy(lhs_xy), _xy(rhs_xy)
    n = l.shape[0]
    F = np.zeros(9, np.float64)
    flags = np.zeros(max(n, 1), np.uint8)
    ni, rounds = ctypes.c_int64(0), ctypes.c_int64(0)
    ok = lib().geo_compute_fundamental(l.ctypes.data, r.ctypes.data, n, int(seed), int(pair),
                                       {"counter": 0, "std": 1}[sampler], F.ctypes.data, flags.ctypes.data,
                                       ctypes.byref(ni), ctypes.byref(rounds))
    return bool(ok), F.reshape(3, 3), flags[:n].copy(), int(rounds.value)
