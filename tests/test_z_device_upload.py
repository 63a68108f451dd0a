"""GPU: images registered from device memory, and the collective replication path on a one-rank
RCCL group.  Kept in a file of its own that sorts after the other GPU tests: it is the only test
that needs a process group."""
import numpy as np
import pytest

from oracle import oracle
from sfm_amd import synth
from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu


@pytest.fixture()
def matcher():
    """A context of its own (not the session's per-engine one): this file is to run last."""
    import sfm_amd
    m = sfm_amd.Matcher(0)
    yield m
    m.close()


def test_images_registered_from_device_memory(matcher):
    """sfm_hip_upload_image_device, and the collective replication path that uses it, on a
    one-rank RCCL group: the all-gather runs on the GPU, the matcher takes the rows from there."""
    import torch
    import torch.distributed as dist
    from sfm_amd.distributed import replicate_images
    imgs = synth.make_image_set(3, 900, 0.6, seed=61)
    images = {200: imgs[0], 201: imgs[1][:411], 202: imgs[2][:0]}
    t = torch.from_numpy(images[200]).to("cuda:0")
    matcher.upload_device(300, t.data_ptr(), t.shape[0])
    matcher.upload(301, images[201])
    got = matcher.match_pairs([(300, 301)], 0.8, True)[0]
    assert_same(got, oracle.match_pair(images[200], images[201], 0.8, True, threads=4))

    started = not dist.is_initialized()
    if started:
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:29536", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        rows = replicate_images(matcher, images, 0, 1, device=torch.device("cuda", 0), method="collective")
    finally:
        if started:
            dist.destroy_process_group()
    assert rows == 900 + 411
    assert [matcher.image_size(i) for i in (200, 201, 202)] == [900, 411, 0]
    for (a, b), got in zip([(200, 201), (201, 202)], matcher.match_pairs([(200, 201), (201, 202)], 0.8, True)):
        assert_same(got, oracle.match_pair(images[a], images[b], 0.8, True, threads=4))


def test_replicate_images_entry_point_one_context(matcher):
    """sfm_hip_replicate_images (one process, RCCL loaded at run time) on the one GPU of the box: a
    communicator of one rank, an in-place all-gather, every image registered from device memory."""
    from sfm_amd.binding import replicate_images
    imgs = synth.make_image_set(4, 700, 0.6, seed=62)
    images = {410: imgs[0], 405: imgs[1][:333], 407: imgs[2][:0], 401: imgs[3][:1]}
    times = replicate_images([matcher], images)
    assert set(times) == {"comm_init_s", "upload_s", "allgather_s", "register_s"} and all(v >= 0 for v in times.values())
    assert [matcher.image_size(i) for i in (401, 405, 407, 410)] == [1, 333, 0, 700]
    pairs = [(401, 405), (405, 410), (407, 410), (405, 407), (401, 410)]
    for (a, b), got in zip(pairs, matcher.match_pairs(pairs, 0.8, True)):
        assert_same(got, oracle.match_pair(images[a], images[b], 0.8, True, threads=4))
    # a second call replaces the images (same communicator), a duplicated id is refused
    images[405] = imgs[1][:500]
    replicate_images([matcher], images)
    assert matcher.image_size(405) == 500
    assert_same(matcher.match_pairs([(405, 410)], 0.8, True)[0], oracle.match_pair(images[405], images[410], 0.8, True, threads=4))
    import sfm_amd
    with pytest.raises(sfm_amd.SfmHipError, match="one GPU per context"):
        replicate_images([matcher, matcher], images)


def test_uploads_that_are_not_waited_for():
    """sfm_hip_upload_image_async: the copy and the preparation of an image are queued, the first call
    that lets the GPU read images waits for them.  Same matches as after blocking uploads, for the
    brute-force and the cascade-hash path, and an image can be replaced that way too."""
    import sfm_amd
    A, B, _ = synth.make_pair(1500, 1700, 600, seed=91)
    C, D, _ = synth.make_pair(900, 1100, 300, seed=92)
    m = sfm_amd.Matcher(0)
    try:
        m.upload(0, A)
        m.upload(1, B)
        m.upload(2, C)
        m.upload(3, D)
        want = m.match_pairs([(0, 1), (2, 3), (0, 3)], 0.8, True)
        m.drop_all()
        for i, d in enumerate((A, B, C, D)):
            m.upload_async(i, d)
        got = m.match_pairs([(0, 1), (2, 3), (0, 3)], 0.8, True)
        for (l, r, d), (wl, wr, wd) in zip(got, want):
            assert l.tolist() == wl.tolist() and r.tolist() == wr.tolist()
            assert d.view(np.uint32).tolist() == wd.view(np.uint32).tolist()
        # replace image 1 by D without waiting, match right away
        m.upload_async(1, D)
        (l, r, d), = m.match_pairs([(0, 1)], 0.8, True)
        (wl, wr, wd) = want[2]
        assert l.tolist() == wl.tolist() and r.tolist() == wr.tolist()
        # the cascade-hash path reads images through the same stream
        P, Q = synth.make_repetitive_set(2, 800, 0.5, 2, seed=8, sigma=0.002)
        m.upload(10, P)
        m.upload(11, Q)
        m.cascade_prepare([(10, 11)])
        cw = m.cascade_match_pairs([(10, 11)], True)
        m.drop(10)
        m.drop(11)
        m.upload_async(10, P)
        m.upload_async(11, Q)
        m.cascade_prepare([(10, 11)])
        cg = m.cascade_match_pairs([(10, 11)], True)
        assert (cg[0][0], cg[0][1].tolist(), cg[0][2].tolist()) == (cw[0][0], cw[0][1].tolist(), cw[0][2].tolist())
    finally:
        m.close()
