"""ctypes front end of the CPU oracle of the cascade-hash matcher (oracle/cascade_oracle.cc).

Test infrastructure only: imported by tests/, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg, never by the sfm_amd package.  Parity with the reference is unpinned for this
matcher; the header of cascade_oracle.cc says what is and is not fixed by the reference's text.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libcascade_oracle.so")

GROUPS, GROUP_BITS, CODE_BITS, DIM = 6, 10, 128, 128

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        l = ctypes.CDLL(_LIB)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        l.ch_set_projection.argtypes = [vp, vp]
        l.ch_set_projection.restype = None
        l.ch_l2_metric.argtypes = [vp, vp]
        l.ch_l2_metric.restype = ctypes.c_float
        l.ch_hamming.argtypes = [vp, vp]
        l.ch_hamming.restype = ctypes.c_int
        l.ch_projection.argtypes = [vp, vp]
        l.ch_projection.restype = None
        l.ch_image_order.argtypes = [vp, i64, vp]
        l.ch_image_order.restype = i64
        l.ch_image_column_sums.argtypes = [vp, i64, vp]
        l.ch_image_column_sums.restype = None
        l.ch_mean.argtypes = [vp, vp, i64, vp]
        l.ch_mean.restype = None
        l.ch_hash.argtypes = [vp, i64, vp, vp, vp]
        l.ch_hash.restype = None
        l.ch_match_pair.argtypes = [vp, i64, vp, i64, vp, ctypes.c_int, vp, vp, vp, vp]
        l.ch_match_pair.restype = i64
        l.ch_match_set.argtypes = [vp, vp, vp, i64, vp, i64, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp]
        l.ch_match_set.restype = i64
        _lib = l
    return _lib


def _desc(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == DIM, a.shape
    return a


def projection():
    """(primary [128, 128], secondary [6, 10, 128]) as the reference draws them."""
    p = np.empty((CODE_BITS, DIM), np.float32)
    s = np.empty((GROUPS, GROUP_BITS, DIM), np.float32)
    lib().ch_projection(p.ctypes.data, s.ctypes.data)
    return p, s


def set_projection(primary=None, secondary=None):
    """Replace the reference's projections in every later oracle call (None, None: restore them)."""
    if primary is None or secondary is None:
        lib().ch_set_projection(None, None)
        return
    p = np.ascontiguousarray(primary, dtype=np.float32)
    s = np.ascontiguousarray(secondary, dtype=np.float32)
    assert p.shape == (CODE_BITS, DIM) and s.shape == (GROUPS, GROUP_BITS, DIM)
    lib().ch_set_projection(p.ctypes.data, s.ctypes.data)


def image_order(pairs):
    """Image ids of a pair list in the order the reference's unordered_set yields them."""
    p = np.ascontiguousarray(sorted(map(tuple, np.asarray(pairs, dtype=np.int64).reshape(-1, 2).tolist())), dtype=np.int64)
    out = np.empty(max(2 * len(p), 1), np.int64)
    n = lib().ch_image_order(p.ctypes.data, len(p), out.ctypes.data)
    return out[:n].copy()


def image_column_sums(desc):
    d = _desc(desc)
    out = np.empty(DIM, np.float32)
    lib().ch_image_column_sums(d.ctypes.data, d.shape[0], out.ctypes.data)
    return out


def mean(images_in_order):
    """The vector the reference subtracts before hashing, from the images in image_order()."""
    ds = [_desc(d) for d in images_in_order]
    ptrs = (ctypes.c_void_p * len(ds))(*[d.ctypes.data for d in ds])
    sizes = np.array([d.shape[0] for d in ds], np.int64)
    out = np.empty(DIM, np.float32)
    lib().ch_mean(ptrs, sizes.ctypes.data, len(ds), out.ctypes.data)
    return out


def hash_image(desc, mean128):
    """(codes [n, 4] uint32, bucket_ids [n, 6] uint16)."""
    d = _desc(desc)
    m = np.ascontiguousarray(mean128, dtype=np.float32)
    codes = np.empty((d.shape[0], 4), np.uint32)
    bids = np.empty((d.shape[0], GROUPS), np.uint16)
    lib().ch_hash(d.ctypes.data, d.shape[0], m.ctypes.data, codes.ctypes.data, bids.ctypes.data)
    return codes, bids


def match_pair(first, second, mean128, cross_check=True):
    """(n_default, lhs_idx, rhs_idx, dist): the reference stores n_default records {0, 0} followed
    by the (lhs_idx, rhs_idx) records."""
    a, b = _desc(first), _desc(second)
    m = np.ascontiguousarray(mean128, dtype=np.float32)
    n = max(a.shape[0], 1)
    ol, orr, od = np.empty(n, np.int64), np.empty(n, np.int64), np.empty(n, np.float32)
    nd = ctypes.c_int64(0)
    c = lib().ch_match_pair(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], m.ctypes.data,
                            int(bool(cross_check)), ctypes.byref(nd), ol.ctypes.data, orr.ctypes.data, od.ctypes.data)
    return int(nd.value), ol[:c].copy(), orr[:c].copy(), od[:c].copy()


def match_set(images, pairs, cross_check=True, threads=1):
    """CascadeHashMatcher::Match over {id: descriptors} and a pair list.
    Returns ({pair: (n_default, lhs_idx, rhs_idx)}, seconds spent hashing, seconds spent matching)."""
    ids = sorted(images)
    ds = [_desc(images[i]) for i in ids]
    ptrs = (ctypes.c_void_p * len(ds))(*[d.ctypes.data for d in ds])
    sizes = np.array([d.shape[0] for d in ds], np.int64)
    ida = np.array(ids, np.int64)
    pl = sorted(map(tuple, np.asarray(pairs, dtype=np.int64).reshape(-1, 2).tolist()))
    p = np.ascontiguousarray(pl, dtype=np.int64).reshape(-1, 2)
    cap = max(int(sum(images[a].shape[0] for a, _ in pl)), 1)
    counts = np.zeros(max(len(pl), 1), np.int64)
    defaults = np.zeros(max(len(pl), 1), np.int64)
    ol, orr = np.empty(cap, np.int64), np.empty(cap, np.int64)
    th, tm = ctypes.c_double(0), ctypes.c_double(0)
    lib().ch_match_set(ida.ctypes.data, ptrs, sizes.ctypes.data, len(ids), p.ctypes.data, len(pl),
                       int(bool(cross_check)), int(threads), counts.ctypes.data, defaults.ctypes.data,
                       ol.ctypes.data, orr.ctypes.data, ctypes.byref(th), ctypes.byref(tm))
    out, w = {}, 0
    for k, pr in enumerate(pl):
        c = int(counts[k])
        out[pr] = (int(defaults[k]), ol[w:w + c].copy(), orr[w:w + c].copy())
        w += c
    return out, th.value, tm.value
