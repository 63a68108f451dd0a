"""ctypes front end of the CPU oracle (oracle/bf_oracle.c).

Test infrastructure only: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg, never by the sfm_amd package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libbf_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libref_metric.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build():
    """Compile the oracle (and oracle/_ref when the reference checkout exists)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def _load():
    if not os.path.exists(_LIB):
        build()
    lib = ctypes.CDLL(_LIB)
    lib.bf_l2_metric.restype = ctypes.c_float
    lib.bf_l2_metric.argtypes = [_f32p, _f32p]
    lib.bf_l2_metric_sq.restype = ctypes.c_float
    lib.bf_l2_metric_sq.argtypes = [_f32p, _f32p]
    lib.bf_match_direction.restype = ctypes.c_int64
    lib.bf_match_direction.argtypes = [_f32p, ctypes.c_int64, _f32p, ctypes.c_int64, ctypes.c_double,
                                       _i64p, _i64p, _f32p, _f32p]
    lib.bf_match_pair.restype = ctypes.c_int64
    lib.bf_match_pair.argtypes = [_f32p, ctypes.c_int64, _f32p, ctypes.c_int64, ctypes.c_double, ctypes.c_int,
                                  _i64p, _i64p, _f32p]
    lib.bf_match_pair_mt.restype = ctypes.c_int64
    lib.bf_match_pair_mt.argtypes = [_f32p, ctypes.c_int64, _f32p, ctypes.c_int64, ctypes.c_double, ctypes.c_int,
                                     ctypes.c_int, _i64p, _i64p, _f32p]
    lib.bf_exhaustive_pairs.restype = ctypes.c_int64
    lib.bf_exhaustive_pairs.argtypes = [_i64p, ctypes.c_int64, _i64p]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _desc(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 128, a.shape
    return a


def _p(a, t):
    return a.ctypes.data_as(t)


def l2_metric(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return np.float32(lib().bf_l2_metric(_p(a, _f32p), _p(b, _f32p)))


def l2_metric_sq(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return np.float32(lib().bf_l2_metric_sq(_p(a, _f32p), _p(b, _f32p)))


def ref_metric_lib():
    """The reference's own metric.hpp as built by oracle/Makefile, or None."""
    if not os.path.exists(_REF):
        return None
    r = ctypes.CDLL(_REF)
    r.ref_L2_metric.restype = ctypes.c_float
    r.ref_L2_metric.argtypes = [_f32p, _f32p]
    if hasattr(r, "ref_HammingDistance128"):
        r.ref_HammingDistance128.restype = ctypes.c_int
        r.ref_HammingDistance128.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    return r


def match_direction(lhs, rhs, ratio=0.8):
    """(lhs_idx, rhs_idx, min_dist, second_dist) of one direction."""
    lhs, rhs = _desc(lhs), _desc(rhs)
    n = lhs.shape[0]
    ol = np.empty(max(n, 1), np.int64)
    orr = np.empty(max(n, 1), np.int64)
    od = np.empty(max(n, 1), np.float32)
    os_ = np.empty(max(n, 1), np.float32)
    c = lib().bf_match_direction(_p(lhs, _f32p), n, _p(rhs, _f32p), rhs.shape[0], ratio,
                                 _p(ol, _i64p), _p(orr, _i64p), _p(od, _f32p), _p(os_, _f32p))
    return ol[:c].copy(), orr[:c].copy(), od[:c].copy(), os_[:c].copy()


def match_pair(first, second, ratio=0.8, cross_check=True, threads=1):
    """(lhs_idx, rhs_idx, dist) as BruteForceMatcher::Match stores them for one pair."""
    first, second = _desc(first), _desc(second)
    n = first.shape[0]
    ol = np.empty(max(n, 1), np.int64)
    orr = np.empty(max(n, 1), np.int64)
    od = np.empty(max(n, 1), np.float32)
    if threads <= 1:
        c = lib().bf_match_pair(_p(first, _f32p), n, _p(second, _f32p), second.shape[0], ratio,
                                int(bool(cross_check)), _p(ol, _i64p), _p(orr, _i64p), _p(od, _f32p))
    else:
        c = lib().bf_match_pair_mt(_p(first, _f32p), n, _p(second, _f32p), second.shape[0], ratio,
                                   int(bool(cross_check)), int(threads), _p(ol, _i64p), _p(orr, _i64p),
                                   _p(od, _f32p))
    return ol[:c].copy(), orr[:c].copy(), od[:c].copy()


def exhaustive_pairs(ids):
    ids = np.ascontiguousarray(sorted(ids), dtype=np.int64)
    n = len(ids)
    out = np.empty(max(n * (n - 1), 2), np.int64)
    c = lib().bf_exhaustive_pairs(_p(ids, _i64p), n, _p(out, _i64p))
    return out[:2 * c].reshape(c, 2).copy()
