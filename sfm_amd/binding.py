"""ctypes binding of include/sfm_hip.h."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SFM_HIP_LIBRARY names another build of the same library (A/B timing of two builds)
_LIB_PATH = os.environ.get("SFM_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libsfm_hip.so")

ENGINES = {"f32": 0, "f16x3": 1}

EXPORTS = (
    "sfm_hip_device_count", "sfm_hip_ctx_create", "sfm_hip_ctx_destroy", "sfm_hip_last_error",
    "sfm_hip_upload_image", "sfm_hip_upload_image_async", "sfm_hip_drop_image", "sfm_hip_drop_all_images", "sfm_hip_image_size",
    "sfm_hip_match_pairs", "sfm_hip_sync", "sfm_hip_host_alloc", "sfm_hip_host_free",
    "sfm_hip_set_profiling", "sfm_hip_get_stats", "sfm_hip_reset_stats", "sfm_hip_set_engine",
    "sfm_hip_upload_image_device", "sfm_hip_match_pair_noalloc", "sfm_hip_partition_pairs",
    "sfm_hip_replicate_images", "sfm_hip_wait_outputs",
    "sfm_hip_cascade_set_projection", "sfm_hip_cascade_column_sums", "sfm_hip_cascade_hash_images",
    "sfm_hip_cascade_get_hashes", "sfm_hip_cascade_match_pairs", "sfm_hip_cascade_reference_projection",
    "sfm_hip_cascade_image_order", "sfm_hip_cascade_prepare", "sfm_hip_cascade_get_stats",
    "sfm_hip_cascade_reset_stats", "sfm_hip_geometry_verify_pairs", "sfm_hip_geometry_get_stats",
    "sfm_hip_geometry_reset_stats",
)

CASCADE_GROUPS, CASCADE_GROUP_BITS, CASCADE_CODE_BITS = 6, 10, 128


class SfmHipError(RuntimeError):
    pass


class Stats(ctypes.Structure):
    _fields_ = [("pairs", ctypes.c_uint64), ("queries", ctypes.c_uint64), ("rescanned_queries", ctypes.c_uint64),
                ("distance_launches", ctypes.c_uint64), ("distance_ms", ctypes.c_double),
                ("distance_flop", ctypes.c_double)]


class ReplicateTimes(ctypes.Structure):
    _fields_ = [("comm_init_s", ctypes.c_double), ("upload_s", ctypes.c_double), ("allgather_s", ctypes.c_double),
                ("register_s", ctypes.c_double)]


class CascadeStats(ctypes.Structure):
    _fields_ = [("pairs", ctypes.c_uint64), ("images_hashed", ctypes.c_uint64), ("match_launches", ctypes.c_uint64),
                ("match_ms", ctypes.c_double), ("match_bytes", ctypes.c_double),
                ("list_overflows", ctypes.c_uint64), ("workspace_dirty_words", ctypes.c_uint64),
                ("walk_ms", ctypes.c_double), ("nearest_ms", ctypes.c_double), ("fallback_ms", ctypes.c_double),
                ("compact_ms", ctypes.c_double), ("walk_candidates", ctypes.c_uint64), ("close_pairs", ctypes.c_uint64),
                ("exact_distances", ctypes.c_uint64)]


class GeometryStats(ctypes.Structure):
    _fields_ = [("calls", ctypes.c_uint64), ("pairs", ctypes.c_uint64), ("timed_launches", ctypes.c_uint64),
                ("rounds", ctypes.c_uint64), ("match_rounds", ctypes.c_uint64),
                ("fit_ms", ctypes.c_double), ("count_ms", ctypes.c_double)]


def library_path():
    return _LIB_PATH


_lib = None


def library():
    """Load libsfm_hip.so.  Raises if it has not been built: there is no other path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise SfmHipError("%s is missing: run `python -m sfm_amd.build` (needs hipcc)" % _LIB_PATH)
    # When torch is part of the process its bundled HIP runtime must be the one
    # both sides use; importing it first makes the loader bind ours to it.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = ctypes.CDLL(_LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    lib.sfm_hip_device_count.argtypes = [ctypes.POINTER(ctypes.c_int)]
    lib.sfm_hip_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    lib.sfm_hip_ctx_destroy.argtypes = [vp]
    lib.sfm_hip_ctx_destroy.restype = None
    lib.sfm_hip_last_error.argtypes = [vp]
    lib.sfm_hip_last_error.restype = ctypes.c_char_p
    lib.sfm_hip_upload_image.argtypes = [vp, i64, vp, i32]
    lib.sfm_hip_upload_image_async.argtypes = [vp, i64, vp, i32]
    lib.sfm_hip_upload_image_device.argtypes = [vp, ctypes.c_int64, vp, ctypes.c_int32]
    lib.sfm_hip_drop_image.argtypes = [vp, i64]
    lib.sfm_hip_drop_all_images.argtypes = [vp]
    lib.sfm_hip_image_size.argtypes = [vp, i64, ctypes.POINTER(i32)]
    lib.sfm_hip_match_pairs.argtypes = [vp, vp, i32, ctypes.c_double, ctypes.c_int, vp, vp, vp, vp, i64]
    lib.sfm_hip_match_pair_noalloc.argtypes = [vp, i64, i64, ctypes.c_double, ctypes.c_int, vp, vp, vp, vp]
    lib.sfm_hip_partition_pairs.argtypes = [vp, i32, vp, vp, i32, i32, vp, vp]
    lib.sfm_hip_replicate_images.argtypes = [vp, i32, vp, vp, vp, i32, ctypes.POINTER(ReplicateTimes)]
    lib.sfm_hip_wait_outputs.argtypes = [vp, vp, ctypes.c_size_t]
    lib.sfm_hip_sync.argtypes = [vp]
    lib.sfm_hip_host_alloc.argtypes = [ctypes.c_size_t]
    lib.sfm_hip_host_alloc.restype = vp
    lib.sfm_hip_host_free.argtypes = [vp]
    lib.sfm_hip_host_free.restype = None
    lib.sfm_hip_set_profiling.argtypes = [vp, ctypes.c_int]
    lib.sfm_hip_set_engine.argtypes = [vp, ctypes.c_int]
    lib.sfm_hip_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    lib.sfm_hip_reset_stats.argtypes = [vp]
    lib.sfm_hip_cascade_set_projection.argtypes = [vp, vp, vp]
    lib.sfm_hip_cascade_column_sums.argtypes = [vp, vp, i32, vp]
    lib.sfm_hip_cascade_hash_images.argtypes = [vp, vp, i32, vp]
    lib.sfm_hip_cascade_get_hashes.argtypes = [vp, i64, vp, vp]
    lib.sfm_hip_cascade_match_pairs.argtypes = [vp, vp, i32, ctypes.c_int, vp, vp, vp, vp, vp, i64]
    lib.sfm_hip_cascade_reference_projection.argtypes = [vp, vp]
    lib.sfm_hip_cascade_image_order.argtypes = [vp, i32, vp, ctypes.POINTER(i32)]
    lib.sfm_hip_cascade_prepare.argtypes = [vp, vp, i32, vp]
    lib.sfm_hip_cascade_get_stats.argtypes = [vp, ctypes.POINTER(CascadeStats)]
    lib.sfm_hip_cascade_reset_stats.argtypes = [vp]
    lib.sfm_hip_geometry_verify_pairs.argtypes = [vp, i32, vp, vp, vp, ctypes.c_uint64, i64, vp, vp, vp, vp]
    lib.sfm_hip_geometry_get_stats.argtypes = [vp, ctypes.POINTER(GeometryStats)]
    lib.sfm_hip_geometry_reset_stats.argtypes = [vp]
    _lib = lib
    return lib


def partition_pairs(pairs, image_rows, n_shards):
    """sfm_hip_partition_pairs: (order, shard_begin).  pairs [n, 2] int64; image_rows {id: rows}.
    Shard s owns the pairs order[shard_begin[s]:shard_begin[s + 1]] (indices into `pairs`)."""
    p = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
    ids = np.ascontiguousarray(sorted(int(i) for i in image_rows), dtype=np.int64)
    rows = np.ascontiguousarray([int(image_rows[int(i)]) for i in ids], dtype=np.int32)
    order = np.empty(max(len(p), 1), np.int32)
    begin = np.empty(int(n_shards) + 1, np.int32)
    rc = library().sfm_hip_partition_pairs(p.ctypes.data, len(p), ids.ctypes.data, rows.ctypes.data, len(ids),
                                           int(n_shards), order.ctypes.data, begin.ctypes.data)
    if rc != 0:
        raise SfmHipError("sfm_hip_partition_pairs failed (%d): a pair names an image without a row count, "
                          "an id appears twice, or a count is negative" % rc)
    return order[:len(p)], begin


def replicate_images(matchers, images):
    """sfm_hip_replicate_images: every Matcher of the list (one per GPU, this process) gets every
    image of {id: float32 [n, 128]}; each image crosses PCIe once, the rest is one RCCL all-gather.
    Returns where the time went."""
    ids = sorted(int(i) for i in images)
    arrays = [np.ascontiguousarray(images[i], dtype=np.float32).reshape(-1, 128) for i in ids]
    id_arr = np.ascontiguousarray(ids, dtype=np.int64)
    rows = np.ascontiguousarray([len(a) for a in arrays], dtype=np.int32)
    ptrs = (ctypes.c_void_p * max(len(ids), 1))(*[a.ctypes.data for a in arrays])
    handles = (ctypes.c_void_p * len(matchers))(*[m._h.value for m in matchers])
    times = ReplicateTimes()
    rc = library().sfm_hip_replicate_images(handles, len(matchers), id_arr.ctypes.data, ptrs, rows.ctypes.data, len(ids),
                                            ctypes.byref(times))
    if rc != 0:
        raise SfmHipError("sfm_hip_replicate_images failed (%d): %s"
                          % (rc, library().sfm_hip_last_error(matchers[0]._h).decode()))
    return {f: getattr(times, f) for f, _ in ReplicateTimes._fields_}


def cascade_reference_projection():
    """(primary [128, 128], secondary [6, 10, 128]): the matrices of cascade_hash_matcher.cc:127-146."""
    p = np.empty((CASCADE_CODE_BITS, 128), np.float32)
    q = np.empty((CASCADE_GROUPS, CASCADE_GROUP_BITS, 128), np.float32)
    rc = library().sfm_hip_cascade_reference_projection(p.ctypes.data, q.ctypes.data)
    if rc != 0:
        raise SfmHipError("sfm_hip_cascade_reference_projection failed (%d)" % rc)
    return p, q


def _sorted_pairs(pairs):
    """The pair list as the reference walks it: a std::set of (first, second), ascending."""
    p = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
    if len(p) == 0:
        return p
    return np.ascontiguousarray(np.unique(p, axis=0))     # unique rows, in lexicographic order


def cascade_image_order(pairs):
    """Image ids of a pair list in the order the reference adds their shares to the mean."""
    p = _sorted_pairs(pairs)
    out = np.empty(max(2 * len(p), 1), np.int64)
    n = ctypes.c_int32(0)
    rc = library().sfm_hip_cascade_image_order(p.ctypes.data, len(p), out.ctypes.data, ctypes.byref(n))
    if rc != 0:
        raise SfmHipError("sfm_hip_cascade_image_order failed (%d)" % rc)
    return out[:n.value].copy()


class _Pinned:
    """A pinned host array owned by the library's allocator."""

    def __init__(self, lib, shape, dtype):
        self._lib = lib
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = lib.sfm_hip_host_alloc(max(n, 1))
        if not self.ptr:
            raise SfmHipError("pinned allocation of %d bytes failed" % n)
        buf = (ctypes.c_char * max(n, 1)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            self._lib.sfm_hip_host_free(self.ptr)
            self.ptr = None


class Matcher:
    """One context on one GPU.  Mirrors the C ABI one to one."""

    def __init__(self, device=0):
        self._lib = library()
        h = ctypes.c_void_p()
        rc = self._lib.sfm_hip_ctx_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise SfmHipError("sfm_hip_ctx_create(%d) failed (%d): %s"
                              % (device, rc, self._lib.sfm_hip_last_error(None).decode()))
        self._h = h
        self._bufs = None
        self._bufs_shape = None

    def close(self):
        if getattr(self, "_h", None):
            if self._bufs:
                for b in self._bufs:
                    b.free()
                self._bufs = None
            if getattr(self, "_cbufs", None):
                for b in self._cbufs:
                    b.free()
                self._cbufs = None
            self._lib.sfm_hip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise SfmHipError("%s failed (%d): %s" % (what, rc, self._lib.sfm_hip_last_error(self._h).decode()))

    def upload(self, image_id, descriptors):
        d = np.ascontiguousarray(descriptors, dtype=np.float32)
        if d.ndim != 2 or d.shape[1] != 128:
            raise ValueError("descriptors must be [n, 128]")
        self._check(self._lib.sfm_hip_upload_image(self._h, int(image_id), d.ctypes.data, d.shape[0]),
                    "sfm_hip_upload_image")

    def upload_async(self, image_id, descriptors):
        """sfm_hip_upload_image_async: queued, not waited for.  The array is kept alive here until the
        next sync()."""
        d = np.ascontiguousarray(descriptors, dtype=np.float32)
        if d.ndim != 2 or d.shape[1] != 128:
            raise ValueError("descriptors must be [n, 128]")
        self._check(self._lib.sfm_hip_upload_image_async(self._h, int(image_id), d.ctypes.data, d.shape[0]),
                    "sfm_hip_upload_image_async")
        if not hasattr(self, "_in_flight"):
            self._in_flight = []
        self._in_flight.append(d)

    def upload_device(self, image_id, device_ptr, n):
        """Register n descriptors (rows of 128 floats) that already lie in this GPU's memory."""
        self._check(self._lib.sfm_hip_upload_image_device(self._h, int(image_id), int(device_ptr), int(n)),
                    "sfm_hip_upload_image_device")

    def drop(self, image_id):
        self._check(self._lib.sfm_hip_drop_image(self._h, int(image_id)), "sfm_hip_drop_image")

    def drop_all(self):
        self._check(self._lib.sfm_hip_drop_all_images(self._h), "sfm_hip_drop_all_images")

    def image_size(self, image_id):
        n = ctypes.c_int32()
        self._check(self._lib.sfm_hip_image_size(self._h, int(image_id), ctypes.byref(n)), "sfm_hip_image_size")
        return n.value

    def _outputs(self, n_pairs, stride):
        # Two sets, used in turn: the library overlaps consecutive calls, and two calls in
        # flight must not write the same host buffers.  A set is kept while the stride stays
        # and the pair count does not grow (a shorter call uses the leading rows), so arrays
        # handed out earlier stay valid across a job's short last call.
        if self._bufs is None or self._bufs_shape[1] != stride or self._bufs_shape[0] < n_pairs:
            if self._bufs:
                self.sync()
                for b in self._bufs:
                    b.free()
            shape = (n_pairs, stride)
            self._bufs = [_Pinned(self._lib, s, t) for _ in range(2)
                          for s, t in (((n_pairs,), np.int32), (shape, np.int32), (shape, np.int32), (shape, np.float32))]
            self._bufs_shape = shape
            self._turn = 0
        self._turn ^= 1
        return self._bufs[4 * self._turn:4 * self._turn + 4]

    def match_pairs_async(self, pairs, ratio=0.8, cross_check=True, stride=None, want_dist=True):
        """Queue the pairs; returns the pinned output arrays (valid after sync()).  want_dist=False
        passes out_dist = NULL, as the C++ matcher class does (the reference's Match record holds no
        distance): the distance array returned is then not written."""
        p = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
        if stride is None:
            stride = max([self.image_size(i) for i in set(p[:, 0].tolist())] + [1])
        cnt, lhs, rhs, dist = self._outputs(p.shape[0], int(stride))
        self._check(self._lib.sfm_hip_match_pairs(self._h, p.ctypes.data, p.shape[0], float(ratio),
                                                  int(bool(cross_check)), cnt.ptr, lhs.ptr, rhs.ptr,
                                                  dist.ptr if want_dist else None, int(stride)), "sfm_hip_match_pairs")
        n = p.shape[0]
        return cnt.array[:n], lhs.array[:n], rhs.array[:n], dist.array[:n]

    def sync(self):
        self._check(self._lib.sfm_hip_sync(self._h), "sfm_hip_sync")
        self._in_flight = []

    def wait_outputs(self, array):
        """Wait until no queued call still has to write `array` (a pinned output array, or a view of
        one, as match_pairs_async returns them); calls that write other memory keep running."""
        a = np.asarray(array)
        self._check(self._lib.sfm_hip_wait_outputs(self._h, a.ctypes.data, a.nbytes), "sfm_hip_wait_outputs")

    def pinned(self, shape, dtype):
        """A pinned host array from the library's allocator (free() it, or leave it to the process)."""
        return _Pinned(self._lib, shape, dtype)

    def match_pair_noalloc_async(self, first, second, ratio, cross_check, cnt, lhs, rhs, dist=None):
        """sfm_hip_match_pair_noalloc on caller-provided _Pinned outputs (valid after sync())."""
        self._check(self._lib.sfm_hip_match_pair_noalloc(self._h, int(first), int(second), float(ratio),
                                                         int(bool(cross_check)), cnt.ptr, lhs.ptr, rhs.ptr,
                                                         dist.ptr if dist is not None else None),
                    "sfm_hip_match_pair_noalloc")

    def match_pairs(self, pairs, ratio=0.8, cross_check=True):
        """[(lhs_idx int64, rhs_idx int64, dist float32)] per pair, as the reference stores them."""
        cnt, lhs, rhs, dist = self.match_pairs_async(pairs, ratio, cross_check)
        self.sync()
        out = []
        for k in range(cnt.shape[0]):
            c = int(cnt[k])
            out.append((lhs[k, :c].astype(np.int64), rhs[k, :c].astype(np.int64), dist[k, :c].copy()))
        return out

    # ---- cascade-hash matcher -------------------------------------------------------------
    def cascade_set_projection(self, primary, secondary):
        p = np.ascontiguousarray(primary, dtype=np.float32)
        q = np.ascontiguousarray(secondary, dtype=np.float32)
        if p.shape != (CASCADE_CODE_BITS, 128) or q.shape != (CASCADE_GROUPS, CASCADE_GROUP_BITS, 128):
            raise ValueError("projection matrices must be [128, 128] and [6, 10, 128]")
        self._check(self._lib.sfm_hip_cascade_set_projection(self._h, p.ctypes.data, q.ctypes.data),
                    "sfm_hip_cascade_set_projection")

    def cascade_column_sums(self, image_ids):
        ids = np.ascontiguousarray(image_ids, dtype=np.int64)
        out = np.empty((len(ids), 128), np.float32)
        self._check(self._lib.sfm_hip_cascade_column_sums(self._h, ids.ctypes.data, len(ids), out.ctypes.data),
                    "sfm_hip_cascade_column_sums")
        return out

    def cascade_hash_images(self, image_ids, mean128):
        ids = np.ascontiguousarray(image_ids, dtype=np.int64)
        m = np.ascontiguousarray(mean128, dtype=np.float32)
        if m.shape != (128,):
            raise ValueError("mean must have 128 entries")
        self._check(self._lib.sfm_hip_cascade_hash_images(self._h, ids.ctypes.data, len(ids), m.ctypes.data),
                    "sfm_hip_cascade_hash_images")

    def cascade_get_hashes(self, image_id):
        """(codes [n, 4] uint32, bucket_ids [n, 6] uint16)"""
        n = self.image_size(image_id)
        codes = np.empty((n, 4), np.uint32)
        bids = np.empty((n, CASCADE_GROUPS), np.uint16)
        self._check(self._lib.sfm_hip_cascade_get_hashes(self._h, int(image_id), codes.ctypes.data, bids.ctypes.data),
                    "sfm_hip_cascade_get_hashes")
        return codes, bids

    def cascade_prepare(self, pairs):
        """Projection, mean and hashes for the images of a pair list; returns the mean."""
        p = _sorted_pairs(pairs)
        mean = np.empty(128, np.float32)
        self._check(self._lib.sfm_hip_cascade_prepare(self._h, p.ctypes.data, len(p), mean.ctypes.data),
                    "sfm_hip_cascade_prepare")
        return mean

    def cascade_match_pairs_async(self, pairs, cross_check=True, stride=None):
        """Queue the pairs; returns pinned (defaults, counts, lhs, rhs, dist), valid after sync()."""
        p = np.ascontiguousarray(pairs, dtype=np.int64).reshape(-1, 2)
        if stride is None:
            stride = max([self.image_size(i) for i in set(p[:, 0].tolist())] + [1])
        shape = (p.shape[0], int(stride))
        if getattr(self, "_cbufs_shape", None) != shape:
            if getattr(self, "_cbufs", None):
                self.sync()
                for b in self._cbufs:
                    b.free()
            self._cbufs = [_Pinned(self._lib, s, t) for _ in range(2)
                           for s, t in (((shape[0],), np.int32), ((shape[0],), np.int32), (shape, np.int32),
                                        (shape, np.int32), (shape, np.float32))]
            self._cbufs_shape = shape
            self._cturn = 0
        self._cturn ^= 1
        dfl, cnt, lhs, rhs, dist = self._cbufs[5 * self._cturn:5 * self._cturn + 5]
        self._check(self._lib.sfm_hip_cascade_match_pairs(self._h, p.ctypes.data, p.shape[0], int(bool(cross_check)),
                                                          dfl.ptr, cnt.ptr, lhs.ptr, rhs.ptr, dist.ptr, int(stride)),
                    "sfm_hip_cascade_match_pairs")
        return dfl.array, cnt.array, lhs.array, rhs.array, dist.array

    def cascade_match_pairs(self, pairs, cross_check=True):
        """[(n_default, lhs_idx int64, rhs_idx int64, dist float32)] per pair: the reference stores
        n_default records {0, 0} followed by the (lhs_idx, rhs_idx) records."""
        dfl, cnt, lhs, rhs, dist = self.cascade_match_pairs_async(pairs, cross_check)
        self.sync()
        out = []
        for k in range(cnt.shape[0]):
            c = int(cnt[k])
            out.append((int(dfl[k]), lhs[k, :c].astype(np.int64), rhs[k, :c].astype(np.int64), dist[k, :c].copy()))
        return out

    def cascade_stats(self):
        s = CascadeStats()
        self._check(self._lib.sfm_hip_cascade_get_stats(self._h, ctypes.byref(s)), "sfm_hip_cascade_get_stats")
        return {f: getattr(s, f) for f, _ in CascadeStats._fields_}

    def cascade_reset_stats(self):
        self._check(self._lib.sfm_hip_cascade_reset_stats(self._h), "sfm_hip_cascade_reset_stats")

    # ---- geometric verification -----------------------------------------------------------
    def geometry_verify_flat(self, offsets, lhs_xy, rhs_xy, seed=0, first_pair_index=0):
        """The C call on flat arrays: offsets int64 [n_pairs + 1], positions float64 [total, 2].
        Returns (F [n_pairs, 9], flags uint8 [total], counts int32 [n_pairs], rounds int32 [n_pairs])."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        lhs = np.ascontiguousarray(lhs_xy, dtype=np.float64)
        rhs = np.ascontiguousarray(rhs_xy, dtype=np.float64)
        n, total = len(offsets) - 1, int(offsets[-1])
        F = np.zeros((n, 9), np.float64)
        flags = np.zeros(max(total, 1), np.uint8)
        counts = np.zeros(max(n, 1), np.int32)
        rounds = np.zeros(max(n, 1), np.int32)
        self._check(self._lib.sfm_hip_geometry_verify_pairs(self._h, n, offsets.ctypes.data, lhs.ctypes.data,
                                                            rhs.ctypes.data, int(seed), int(first_pair_index),
                                                            F.ctypes.data, flags.ctypes.data, counts.ctypes.data,
                                                            rounds.ctypes.data), "sfm_hip_geometry_verify_pairs")
        return F, flags[:total], counts[:n], rounds[:n]

    def geometry_verify_pairs(self, matches, seed=0, first_pair_index=0):
        """matches: [(lhs_xy [n, 2], rhs_xy [n, 2])] per pair.  Returns per pair
        (ok, F [3, 3], inlier flags uint8 [n], rounds) as ComputeFundamentalMatrix decides them."""
        offsets = np.zeros(len(matches) + 1, np.int64)
        offsets[1:] = np.cumsum([len(l) for l, _ in matches])
        lhs = np.concatenate([np.asarray(l, np.float64).reshape(-1, 2) for l, _ in matches] or [np.zeros((0, 2))])
        rhs = np.concatenate([np.asarray(r, np.float64).reshape(-1, 2) for _, r in matches] or [np.zeros((0, 2))])
        F, flags, counts, rounds = self.geometry_verify_flat(offsets, lhs, rhs, seed, first_pair_index)
        return [(int(counts[k]) > 30, F[k].reshape(3, 3).copy(), flags[offsets[k]:offsets[k + 1]].copy(), int(rounds[k]))
                for k in range(len(matches))]

    def geometry_stats(self):
        s = GeometryStats()
        self._check(self._lib.sfm_hip_geometry_get_stats(self._h, ctypes.byref(s)), "sfm_hip_geometry_get_stats")
        return {f: getattr(s, f) for f, _ in GeometryStats._fields_}

    def geometry_reset_stats(self):
        self._check(self._lib.sfm_hip_geometry_reset_stats(self._h), "sfm_hip_geometry_reset_stats")

    def set_engine(self, name):
        """'f32' (fp32 matrix instructions) or 'f16x3' (split fp16 products): include/sfm_hip.h."""
        self._check(self._lib.sfm_hip_set_engine(self._h, ENGINES[name]), "sfm_hip_set_engine")

    def set_profiling(self, on):
        """False/True, or 2: per-kernel events and work counters in cascade-hash batches as well."""
        self._check(self._lib.sfm_hip_set_profiling(self._h, 2 if on == 2 and on is not True else int(bool(on))),
                    "sfm_hip_set_profiling")

    def stats(self):
        s = Stats()
        self._check(self._lib.sfm_hip_get_stats(self._h, ctypes.byref(s)), "sfm_hip_get_stats")
        return {f: getattr(s, f) for f, _ in Stats._fields_}

    def reset_stats(self):
        self._check(self._lib.sfm_hip_reset_stats(self._h), "sfm_hip_reset_stats")
