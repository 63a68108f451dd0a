"""sfm_hip_match_pair_noalloc (the entry point that allocates nothing) against sfm_hip_match_pairs.

Runs last: the fixed work space it uses lives in the library's GPU code for the life of the
process, and a failure here must not hide other tests behind `pytest -x`.
"""
import numpy as np
import pytest

from sfm_amd import synth

pytestmark = pytest.mark.gpu

MAX_ROWS = 8192


@pytest.fixture(scope="module", params=["f32", "f16x3"])
def matcher(request):
    """A context of this module's own, per distance engine (the session-wide one would pull these
    tests forward in the collection order, next to the other tests that share it)."""
    import sfm_amd
    m = sfm_amd.Matcher(0)
    m.set_engine(request.param)
    yield m
    m.close()


def _both(m, a, b, ratio, cross, bufs):
    """(reference entry point, allocation-free entry point) results for the uploaded pair (a, b)."""
    (el, er, ed), = m.match_pairs([(a, b)], ratio, cross)
    cnt, lhs, rhs, dist = bufs
    cnt.array[:] = -7
    lhs.array[:] = -7
    rhs.array[:] = -7
    dist.array[:] = -7.0
    m.match_pair_noalloc_async(a, b, ratio, cross, cnt, lhs, rhs, dist)
    m.sync()
    c = int(cnt.array[0])
    return (el, er, ed), (lhs.array[:c].astype(np.int64), rhs.array[:c].astype(np.int64), dist.array[:c].copy(), c)


def _same(expected, got, n_first):
    el, er, ed = expected
    l, r, d, c = got
    assert 0 <= c <= n_first
    assert l.tolist() == el.tolist() and r.tolist() == er.tolist()
    assert d.view(np.uint32).tolist() == ed.view(np.uint32).tolist()     # bit for bit


def test_two_calls_in_a_row_second_smaller(matcher):
    m = matcher
    bufs = [m.pinned((1,), np.int32), m.pinned((MAX_ROWS,), np.int32), m.pinned((MAX_ROWS,), np.int32),
            m.pinned((MAX_ROWS,), np.float32)]
    try:
        A, B, _ = synth.make_pair(8192, 8192, 4096, seed=31)
        C, D, _ = synth.make_pair(700, 900, 300, seed=32)
        m.upload(600, A)
        m.upload(601, B)
        m.upload(602, C)
        m.upload(603, D)
        for cross in (True, False):
            # the largest pair the entry point takes, then a smaller one on the same work space
            e1, g1 = _both(m, 600, 601, 0.8, cross, bufs)
            _same(e1, g1, 8192)
            assert g1[3] >= 4000
            e2, g2 = _both(m, 602, 603, 0.8, cross, bufs)
            _same(e2, g2, 700)
            assert g2[3] >= 250
        # ragged shapes: few rows against many, both ways round
        for a, b in ((602, 601), (600, 603)):
            e, g = _both(m, a, b, 0.8, True, bufs)
            _same(e, g, m.image_size(a))
        # back-to-back calls without a sync in between: the last one's results stand
        m.match_pair_noalloc_async(600, 601, 0.8, True, *bufs)
        m.match_pair_noalloc_async(602, 603, 0.8, True, *bufs)
        m.sync()
        c = int(bufs[0].array[0])
        (el, er, ed), = m.match_pairs([(602, 603)], 0.8, True)
        assert c == len(el) and bufs[1].array[:c].tolist() == el.tolist() and bufs[2].array[:c].tolist() == er.tolist()
    finally:
        m.sync()
        for i in (600, 601, 602, 603):
            m.drop(i)
        for b in bufs:
            b.free()


def test_calls_allocate_nothing_and_are_seen_by_wait_outputs(matcher):
    """Free device memory is the same after a run of calls as before it (the work space is part of the
    code object, every stream and event was made with the context), and sfm_hip_wait_outputs on a
    call's count waits for that call."""
    import torch
    m = matcher
    bufs = [m.pinned((1,), np.int32), m.pinned((MAX_ROWS,), np.int32), m.pinned((MAX_ROWS,), np.int32),
            m.pinned((MAX_ROWS,), np.float32)]
    try:
        A, B, _ = synth.make_pair(4096, 5000, 2000, seed=35)
        m.upload(630, A)
        m.upload(631, B)
        (el, er, ed), = m.match_pairs([(630, 631)], 0.8, True)
        m.match_pair_noalloc_async(630, 631, 0.8, True, *bufs)     # first use of everything involved
        m.sync()
        torch.cuda.mem_get_info(0)                                 # whatever torch sets up on first use
        free_before = torch.cuda.mem_get_info(0)[0]
        for k in range(12):
            bufs[0].array[0] = -7
            m.match_pair_noalloc_async(630, 631, 0.8, k % 2 == 0, *bufs)
            if k % 4 == 0:
                m.wait_outputs(bufs[0].array)
                assert int(bufs[0].array[0]) == len(el)            # k even: cross-check on
        m.sync()
        assert torch.cuda.mem_get_info(0)[0] == free_before
        m.match_pair_noalloc_async(630, 631, 0.8, True, *bufs)
        m.wait_outputs(bufs[0].array)
        c = int(bufs[0].array[0])
        assert bufs[1].array[:c].tolist() == el.tolist() and bufs[2].array[:c].tolist() == er.tolist()
        assert bufs[3].array[:c].view(np.uint32).tolist() == ed.view(np.uint32).tolist()
    finally:
        m.sync()
        m.drop(630)
        m.drop(631)
        for b in bufs:
            b.free()


def test_empty_image_and_no_distances(matcher):
    m = matcher
    bufs = [m.pinned((1,), np.int32), m.pinned((MAX_ROWS,), np.int32), m.pinned((MAX_ROWS,), np.int32)]
    try:
        A, B, _ = synth.make_pair(300, 500, 100, seed=33)
        m.upload(610, A)
        m.upload(611, B)
        m.upload(612, np.zeros((0, 128), np.float32))
        for a, b in ((610, 612), (612, 610)):
            bufs[0].array[0] = -7
            m.match_pair_noalloc_async(a, b, 0.8, True, bufs[0], bufs[1], bufs[2], None)
            m.sync()
            assert int(bufs[0].array[0]) == 0
        (el, er, _), = m.match_pairs([(610, 611)], 0.8, True)
        m.match_pair_noalloc_async(610, 611, 0.8, True, bufs[0], bufs[1], bufs[2], None)   # out_dist NULL
        m.sync()
        c = int(bufs[0].array[0])
        assert bufs[1].array[:c].tolist() == el.tolist() and bufs[2].array[:c].tolist() == er.tolist()
    finally:
        m.sync()
        for i in (610, 611, 612):
            m.drop(i)
        for b in bufs:
            b.free()


def test_refusals(matcher):
    import sfm_amd
    m = matcher
    bufs = [m.pinned((1,), np.int32), m.pinned((MAX_ROWS,), np.int32), m.pinned((MAX_ROWS,), np.int32)]
    try:
        m.upload(620, np.ones((MAX_ROWS + 1, 128), np.float32))
        m.upload(621, np.ones((10, 128), np.float32))
        for a, b in ((620, 621), (621, 620)):
            with pytest.raises(sfm_amd.SfmHipError, match="at most 8192"):
                m.match_pair_noalloc_async(a, b, 0.8, True, bufs[0], bufs[1], bufs[2], None)
        with pytest.raises(sfm_amd.SfmHipError, match="not uploaded"):
            m.match_pair_noalloc_async(621, 999999, 0.8, True, bufs[0], bufs[1], bufs[2], None)
        # pageable outputs cannot be written from the GPU and copying them would need staging memory
        pageable = np.zeros(MAX_ROWS, np.int32)

        class Fake:
            ptr = pageable.ctypes.data
        with pytest.raises(sfm_amd.SfmHipError, match="pinned host memory"):
            m.match_pair_noalloc_async(621, 621, 0.8, True, bufs[0], Fake, bufs[2], None)
    finally:
        m.sync()
        m.drop(620)
        m.drop(621)
        for b in bufs:
            b.free()
