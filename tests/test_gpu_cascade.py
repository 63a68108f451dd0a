"""GPU parity of the cascade-hash path (include/sfm_hip.h, sfm_hip_cascade_*) against the CPU oracle
(oracle/cascade_oracle.cc), through the C ABI.  Everything is compared bit for bit: column sums,
mean, hash codes, bucket ids, match indices, the number of default records and distances.

Parity with the reference itself is unpinned for this matcher (no Eigen here to build it, no
fixture in the reference); the oracle's header says which parts of the reference's text it follows
literally and which it has to fix by convention (the order of float sums inside Eigen).
"""
import numpy as np
import pytest

import sfm_amd
from sfm_amd import binding, synth
from oracle import cascade

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def matcher():
    m = sfm_amd.Matcher(0)
    yield m
    # every batch leaves the per-query work space all zeros
    assert m.cascade_stats()["workspace_dirty_words"] == 0
    m.close()


def _upload(m, images):
    m.drop_all()
    for i, d in images.items():
        m.upload(i, d)


def _oracle_mean(images, pairs):
    order = cascade.image_order(pairs)
    return cascade.mean([images[int(i)] for i in order])


def _check_pairs(m, images, pairs, cross):
    got = m.cascade_match_pairs(pairs, cross_check=cross)
    mean = _oracle_mean(images, pairs)
    total = 0
    for (a, b), (nd, l, r, d) in zip(pairs, got):
        end, el, er, ed = cascade.match_pair(images[a], images[b], mean, cross)
        assert nd == end, (a, b, nd, end)
        assert l.tolist() == el.tolist(), (a, b)
        assert r.tolist() == er.tolist(), (a, b)
        assert d.view(np.uint32).tolist() == ed.view(np.uint32).tolist(), (a, b)
        total += len(l)
    return total


def test_known_answer_case(matcher):
    """The hand-derived case of tests/cascade_kat.py through the C ABI: coordinate projections in
    place of the random ones, zero mean."""
    from cascade_kat import build, coordinate_projection
    A, B, expected = build()
    p, q = coordinate_projection()
    _upload(matcher, {0: A, 1: B})
    matcher.cascade_set_projection(p, q)
    matcher.cascade_hash_images([0, 1], np.zeros(128, np.float32))
    cascade.set_projection(p, q)
    try:
        for cross in (False, True):
            (nd, l, r, d), = matcher.cascade_match_pairs([(0, 1)], cross)
            assert (nd, list(zip(l.tolist(), r.tolist()))) == expected[cross]
            _, _, _, od = cascade.match_pair(A, B, np.zeros(128, np.float32), cross)
            assert d.view(np.uint32).tolist() == od.view(np.uint32).tolist()
    finally:
        cascade.set_projection(None, None)
        rp, rq = binding.cascade_reference_projection()
        matcher.cascade_set_projection(rp, rq)


@pytest.mark.parametrize("n", [0, 1, 5, 64, 129, 1000, 8192])
def test_column_sums_bitwise(matcher, n):
    d = synth.rootsift_rows(n, 77 + n)
    _upload(matcher, {3: d})
    got = matcher.cascade_column_sums([3])[0]
    want = cascade.image_column_sums(d)
    assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist()


@pytest.mark.parametrize("sizes", [(700, 900), (1, 64), (65, 0), (4096, 4096)])
def test_prepare_mean_and_hashes_bitwise(matcher, sizes):
    images = {10 + k: synth.rootsift_rows(n, 5 + 13 * k + n) for k, n in enumerate(sizes)}
    _upload(matcher, images)
    pairs = [(10, 11)]
    mean = matcher.cascade_prepare(pairs)
    want_mean = _oracle_mean(images, pairs)
    assert mean.view(np.uint32).tolist() == want_mean.view(np.uint32).tolist()
    for i, d in images.items():
        codes, bids = matcher.cascade_get_hashes(i)
        wc, wb = cascade.hash_image(d, want_mean)
        assert codes.tolist() == wc.tolist()
        assert bids.tolist() == wb.tolist()


@pytest.mark.parametrize("cross", [True, False])
@pytest.mark.parametrize("n,copies,sigma", [(2048, 2, 0.002), (3000, 3, 0.001), (1500, 2, 0.004)])
def test_pair_matches_identical_to_oracle(matcher, n, copies, sigma, cross):
    a, b = synth.make_repetitive_set(2, n, 0.5, copies, seed=n + copies, sigma=sigma)
    images = {0: a, 1: b[: n - 37]}
    _upload(matcher, images)
    matcher.cascade_prepare([(0, 1)])
    assert _check_pairs(matcher, images, [(0, 1)], cross) > 50


def test_set_of_images_all_pairs(matcher):
    imgs = synth.make_repetitive_set(5, 1200, 0.6, 2, seed=3, sigma=0.002)
    images = {2 * k + 1: d[: 1200 - 50 * k] for k, d in enumerate(imgs)}
    ids = sorted(images)
    pairs = [(a, b) for x, a in enumerate(ids) for b in ids[x + 1:]]
    _upload(matcher, images)
    matcher.cascade_prepare(pairs)
    for cross in (True, False):
        assert _check_pairs(matcher, images, pairs, cross) > 100
    # the whole job in one oracle call, as CascadeHashMatcher::Match runs it
    whole, _, _ = cascade.match_set(images, pairs, True)
    got = matcher.cascade_match_pairs(pairs, True)
    for pr, (nd, l, r, _) in zip(pairs, got):
        wnd, wl, wr = whole[pr]
        assert (nd, l.tolist(), r.tolist()) == (wnd, wl.tolist(), wr.tolist())


def test_ties_and_crowded_bins(matcher):
    """Exact duplicates: equal distances are decided by the lower index (pairs are compared as
    (distance, index)), and a Hamming bin is taken whole even when it holds far more than ten."""
    base = synth.rootsift_rows(600, 91)
    a = base.copy()
    b = synth.rootsift_rows(600, 92)
    b[100:103] = a[5]          # three exact copies of one row: distance 0 three times
    b[200:240] = a[7]          # forty copies: bin 0 holds 40 candidates
    b[300] = a[9]              # a single copy: one candidate only, no match (needs two)
    a[20:23] = a[5]            # and several queries that see the same copies
    images = {0: a, 1: b}
    _upload(matcher, images)
    matcher.cascade_prepare([(0, 1)])
    for cross in (False, True):
        _check_pairs(matcher, images, [(0, 1)], cross)
    (nd, l, r, d), = matcher.cascade_match_pairs([(0, 1)], False)
    got = dict(zip(l.tolist(), r.tolist()))
    assert got[5] == 100 and got[20] == 100 and got[7] == 200 and 9 not in got
    assert d[l.tolist().index(5)] == 0.0


def test_all_identical_descriptors(matcher):
    """Every descriptor in one bucket with one hash code: the largest bins there can be."""
    row = synth.rootsift_rows(1, 5)
    images = {0: np.repeat(row, 300, axis=0), 1: np.repeat(row, 200, axis=0)}
    _upload(matcher, images)
    matcher.cascade_prepare([(0, 1)])
    before = matcher.cascade_stats()["list_overflows"]
    for cross in (False, True):
        assert _check_pairs(matcher, images, [(0, 1)], cross) >= 1
    # 60,000 close pairs do not fit the pair's list: the list-free path answered
    st = matcher.cascade_stats()
    assert st["list_overflows"] == before + 2 and st["workspace_dirty_words"] == 0
    # and an ordinary pair right after it, in the same work space
    a, b = synth.make_repetitive_set(2, 700, 0.5, 2, seed=12, sigma=0.002)
    images = {0: a, 1: b}
    _upload(matcher, images)
    matcher.cascade_prepare([(0, 1)])
    assert _check_pairs(matcher, images, [(0, 1)], True) > 20
    assert matcher.cascade_stats()["list_overflows"] == before + 2


def test_empty_and_single_images(matcher):
    d = synth.make_repetitive_set(1, 500, 0.5, 2, seed=8)[0]
    images = {0: d, 1: np.zeros((0, 128), np.float32), 2: d[:1].copy(), 3: d.copy()}
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    _upload(matcher, images)
    matcher.cascade_prepare(pairs)
    for cross in (False, True):
        _check_pairs(matcher, images, pairs, cross)


def test_errors(matcher):
    d = synth.rootsift_rows(100, 1)
    _upload(matcher, {0: d, 1: d})
    with pytest.raises(binding.SfmHipError, match="not been hashed"):
        matcher.cascade_match_pairs([(0, 1)])
    with pytest.raises(binding.SfmHipError, match="not uploaded"):
        matcher.cascade_column_sums([7])
    matcher.cascade_prepare([(0, 1)])
    matcher.upload(1, d)       # a new upload discards the image's hashes
    with pytest.raises(binding.SfmHipError, match="not been hashed"):
        matcher.cascade_match_pairs([(0, 1)])


def test_many_pairs_split_into_batches_and_pageable_outputs(matcher):
    """More pairs than one launch takes (1024), results into ordinary (pageable) host arrays."""
    import ctypes
    imgs = synth.make_repetitive_set(48, 256, 0.7, 2, seed=14, sigma=0.002)
    images = {i: d[: 256 - (i % 5)] for i, d in enumerate(imgs)}
    ids = sorted(images)
    pairs = [(a, b) for x, a in enumerate(ids) for b in ids[x + 1:]]
    assert len(pairs) == 1128
    _upload(matcher, images)
    mean = matcher.cascade_prepare(pairs)
    p = np.ascontiguousarray(pairs, np.int64)
    stride = 256
    dfl = np.full(len(pairs), -7, np.int32)
    cnt = np.full(len(pairs), -7, np.int32)
    lhs = np.full((len(pairs), stride), -7, np.int32)
    rhs = np.full((len(pairs), stride), -7, np.int32)
    dist = np.zeros((len(pairs), stride), np.float32)
    lib = binding.library()
    rc = lib.sfm_hip_cascade_match_pairs(matcher._h, p.ctypes.data, len(pairs), 1, dfl.ctypes.data, cnt.ctypes.data,
                                         lhs.ctypes.data, rhs.ctypes.data, dist.ctypes.data, stride)
    assert rc == 0
    matcher.sync()
    total = 0
    for k, (a, b) in enumerate(pairs):
        end, el, er, ed = cascade.match_pair(images[a], images[b], mean, True)
        c = int(cnt[k])
        assert (int(dfl[k]), lhs[k, :c].tolist(), rhs[k, :c].tolist()) == (end, el.tolist(), er.tolist()), (a, b)
        assert dist[k, :c].view(np.uint32).tolist() == ed.view(np.uint32).tolist()
        total += c
    assert total > 1000
    assert matcher.cascade_stats()["workspace_dirty_words"] == 0


def test_large_and_unequal_images(matcher):
    a, b = synth.make_repetitive_set(2, 20000, 0.4, 2, seed=15, sigma=0.002)
    images = {0: a, 1: b[:7001], 2: a[:300]}
    pairs = [(0, 1), (0, 2), (1, 2)]
    _upload(matcher, images)
    matcher.cascade_prepare(pairs)
    assert _check_pairs(matcher, images, pairs, True) > 50
    assert _check_pairs(matcher, images, [(1, 0), (2, 0)], False) > 50    # first image the smaller one


def test_image_size_limit_of_the_cascade_path(matcher):
    """262,144 descriptors per image at most (the close-pair list lays out one count per 256 records
    and group in LDS); one more is refused with a message, nothing is launched."""
    big = np.zeros((262145, 128), np.float32)
    _upload(matcher, {0: big, 1: big[:10]})
    rp, rq = binding.cascade_reference_projection()
    matcher.cascade_set_projection(rp, rq)
    with pytest.raises(binding.SfmHipError, match="262,144"):
        matcher.cascade_hash_images([0, 1], np.zeros(128, np.float32))
    matcher.drop_all()


def test_refused_hashing_leaves_earlier_hashes_usable(matcher):
    """A call that names a small image and one beyond the size limit is refused as a whole: the
    small image keeps the hashes it had (matching it still gives the oracle's result), and an image
    that never had hashes still has none (matching it is refused, not run on unfilled memory)."""
    a, b = synth.make_repetitive_set(2, 1500, 0.6, 2, seed=19, sigma=0.002)
    big = np.zeros((262145, 128), np.float32)
    images = {0: a, 1: b}
    _upload(matcher, {0: a, 1: b, 2: big, 3: a[:40]})
    pairs = [(0, 1)]
    mean = matcher.cascade_prepare(pairs)
    before = _check_pairs(matcher, images, pairs, True)
    assert before > 20
    for ids in ([0, 2], [2, 0], [3, 2]):
        with pytest.raises(binding.SfmHipError, match="262,144"):
            matcher.cascade_hash_images(ids, mean)
    assert _check_pairs(matcher, images, pairs, True) == before
    with pytest.raises(binding.SfmHipError, match="hash"):
        matcher.cascade_match_pairs([(3, 0)], True)
    matcher.drop_all()


def test_calls_in_flight_that_reuse_output_buffers_land_in_call_order(matcher):
    """Consecutive cascade batches run on two streams in turn.  Two sets of output buffers used in
    turn by eight calls in flight, a big pair (slow) followed by small ones: each set must end up
    holding its LAST call's result."""
    import ctypes
    from sfm_amd.binding import _Pinned
    lib = sfm_amd.library()
    imgs = synth.make_repetitive_set(4, 20000, 0.5, 2, seed=27, sigma=0.002)
    images = {0: imgs[0], 1: imgs[1], 2: imgs[2][:400], 3: imgs[3][:300]}
    _upload(matcher, images)
    calls = [(2, 3), (3, 2), (0, 1), (1, 0), (2, 3), (3, 2), (3, 2), (2, 3)]
    everything = sorted(set(calls))
    matcher.cascade_prepare(everything)
    mean = _oracle_mean(images, everything)
    stride = 20000
    sets = [[_Pinned(lib, s, t) for s, t in (((1,), np.int32), ((1,), np.int32), ((1, stride), np.int32),
                                             ((1, stride), np.int32), ((1, stride), np.float32))] for _ in range(2)]
    try:
        for k, pair in enumerate(calls):
            p = np.array([pair], np.int64)
            dfl, cnt, lhs, rhs, dist = sets[k & 1]
            rc = lib.sfm_hip_cascade_match_pairs(matcher._h, p.ctypes.data, 1, 1, dfl.ptr, cnt.ptr, lhs.ptr, rhs.ptr,
                                                 dist.ptr, stride)
            assert rc == 0, lib.sfm_hip_last_error(matcher._h)
        matcher.sync()
        for which in (0, 1):
            a, b = [p for k, p in enumerate(calls) if (k & 1) == which][-1]
            dfl, cnt, lhs, rhs, dist = sets[which]
            end, el, er, ed = cascade.match_pair(images[a], images[b], mean, True)
            c = int(cnt.array[0])
            assert (int(dfl.array[0]), lhs.array[0, :c].tolist(), rhs.array[0, :c].tolist()) == (end, el.tolist(), er.tolist())
            assert dist.array[0, :c].view(np.uint32).tolist() == ed.view(np.uint32).tolist()
    finally:
        matcher.sync()
        for bufs in sets:
            for b in bufs:
                b.free()
        matcher.drop_all()


def test_wait_outputs_covers_an_earlier_multi_batch_call(matcher):
    """sfm_hip_wait_outputs on a cascade-hash call whose lanes have been taken again since: call 1
    needs two batches (1,128 pairs, one batch on each of the two lanes), call 2 is queued behind it
    into a second set of buffers and takes both lanes again without a host wait.  Waiting for call
    1's counts must still wait for call 1 -- its results are read right afterwards, with call 2
    possibly still running."""
    from sfm_amd.binding import _Pinned
    lib = sfm_amd.library()
    imgs = synth.make_repetitive_set(48, 2048, 0.6, 2, seed=41, sigma=0.002)
    images = {i: d[: 2048 - 7 * (i % 5)] for i, d in enumerate(imgs)}
    ids = sorted(images)
    pairs = [(a, b) for x, a in enumerate(ids) for b in ids[x + 1:]]
    assert len(pairs) == 1128                      # more than one batch of 1,024
    _upload(matcher, images)
    mean = matcher.cascade_prepare(pairs)
    p1 = np.ascontiguousarray(pairs, np.int64)
    p2 = np.ascontiguousarray(pairs[::-1], np.int64)          # call 2: the same pairs in another order
    stride, n = 2048, len(pairs)
    sets = [[_Pinned(lib, s, t) for s, t in (((n,), np.int32), ((n,), np.int32), ((n, stride), np.int32),
                                             ((n, stride), np.int32), ((n, stride), np.float32))] for _ in range(2)]
    try:
        for rep in range(3):
            for bufs in sets:
                bufs[0].array[:] = -7
                bufs[1].array[:] = -7
            for p, bufs in ((p1, sets[0]), (p2, sets[1])):
                rc = lib.sfm_hip_cascade_match_pairs(matcher._h, p.ctypes.data, n, 1, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr,
                                                     bufs[3].ptr, bufs[4].ptr, stride)
                assert rc == 0, lib.sfm_hip_last_error(matcher._h)
            matcher.wait_outputs(sets[0][1].array)
            dfl, cnt = sets[0][0].array.copy(), sets[0][1].array.copy()        # call 1's, as they are NOW
            lhs_now = {k: sets[0][2].array[k, :max(int(cnt[k]), 0)].tolist() for k in (0, 500, 1023, 1024, 1127)}
            assert cnt.min() >= 0 and dfl.min() >= 0, "call 1 had not finished when sfm_hip_wait_outputs returned"
            if rep == 0:
                total = 0
                for k in list(range(0, n, 37)) + list(range(1024, n, 9)):
                    a, b = pairs[k]
                    end, el, er, ed = cascade.match_pair(images[a], images[b], mean, True)
                    c = int(cnt[k])
                    assert (int(dfl[k]), sets[0][2].array[k, :c].tolist(), sets[0][3].array[k, :c].tolist()) == \
                        (end, el.tolist(), er.tolist()), (a, b)
                    assert sets[0][4].array[k, :c].view(np.uint32).tolist() == ed.view(np.uint32).tolist()
                    total += c
                assert total > 500
                first = (dfl, cnt, lhs_now)
            else:     # later rounds: the same answer, read at the same moment
                assert np.array_equal(first[0], dfl) and np.array_equal(first[1], cnt) and first[2] == lhs_now
            matcher.sync()
            # call 2 wrote the same pairs in reverse order into the other set
            assert np.array_equal(sets[1][1].array[::-1], cnt) and np.array_equal(sets[1][0].array[::-1], dfl)
        assert matcher.cascade_stats()["workspace_dirty_words"] == 0
    finally:
        matcher.sync()
        for bufs in sets:
            for b in bufs:
                b.free()
        matcher.drop_all()


def test_per_kernel_statistics_leave_the_matches_alone(matcher):
    """sfm_hip_set_profiling(ctx, 2): events between the kernels of a batch and work counters inside
    them.  Same matches; the walk reports at least as many candidates as close pairs, and every
    match comes from a close pair."""
    a, b = synth.make_repetitive_set(2, 2048, 0.5, 2, seed=77, sigma=0.002)
    images = {0: a, 1: b}
    _upload(matcher, images)
    matcher.cascade_prepare([(0, 1)])
    plain = matcher.cascade_match_pairs([(0, 1)], True)
    matcher.set_profiling(2)
    matcher.cascade_reset_stats()
    timed = matcher.cascade_match_pairs([(0, 1)], True)
    st = matcher.cascade_stats()
    matcher.set_profiling(True)
    matcher.cascade_reset_stats()
    light = matcher.cascade_match_pairs([(0, 1)], True)
    st_light = matcher.cascade_stats()
    matcher.set_profiling(False)
    for got in (timed, light):
        assert (got[0][0], got[0][1].tolist(), got[0][2].tolist()) == (plain[0][0], plain[0][1].tolist(), plain[0][2].tolist())
    assert st["match_launches"] == 1 and st["walk_ms"] > 0 and st["nearest_ms"] > 0 and st["compact_ms"] > 0
    assert st["walk_ms"] + st["nearest_ms"] + st["fallback_ms"] + st["compact_ms"] <= st["match_ms"] * 1.001
    assert st["walk_candidates"] >= st["close_pairs"] >= st["exact_distances"] >= len(plain[0][1]) > 50
    # bucket sharing is symmetric: the count equals the sum over groups and buckets of |A_b| * |B_b|
    _, bids_a = matcher.cascade_get_hashes(0)
    _, bids_b = matcher.cascade_get_hashes(1)
    want = sum(int(np.dot(np.bincount(bids_a[:, g], minlength=1024), np.bincount(bids_b[:, g], minlength=1024)))
               for g in range(binding.CASCADE_GROUPS))
    assert st["walk_candidates"] == want
    assert st_light["match_launches"] == 1 and st_light["walk_ms"] == 0 and st_light["walk_candidates"] == 0
