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
