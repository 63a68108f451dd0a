"""CPU-only: the multi-rank path (static pair sharding + gather) over gloo with two processes."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT
from sfm_amd.distributed import exhaustive_pairs, images_of, partition, shard_bounds


def test_shard_bounds_tile_the_range():
    for n in (0, 1, 7, 4950, 124750):
        for world in (1, 2, 3, 8):
            edges = [shard_bounds(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def test_exhaustive_pairs_match_the_oracle():
    from oracle import oracle
    ids = [9, 2, 5, 40, 41]
    assert exhaustive_pairs(ids).tolist() == oracle.exhaustive_pairs(ids).tolist()
    assert len(exhaustive_pairs(range(100))) == 4950


def test_partition_is_balanced_and_local_on_a_ragged_set():
    """Shards of equal cost (rows x rows per pair) to within 2 %, and each names about 2 n / sqrt(G)
    images or fewer instead of all n."""
    rng = np.random.default_rng(5)
    for n, world in ((500, 8), (500, 4), (200, 8), (64, 2), (300, 3)):
        rows = {1000 + 3 * i: int(rng.integers(500, 16384)) for i in range(n)}
        pairs = exhaustive_pairs(rows)
        shards = partition(pairs, rows, world)
        assert sorted(np.concatenate(shards).tolist()) == list(range(len(pairs)))      # a partition
        cost = np.array([rows[a] * rows[b] for a, b in pairs.tolist()], np.float64)
        per = np.array([cost[s].sum() for s in shards])
        assert per.max() / per.mean() - 1 <= 0.02, (n, world, per)
        if world >= 4:
            bound = 2 * n / np.sqrt(world)
            assert max(len(images_of(pairs, s)) for s in shards) <= bound, (n, world)
        # the same arguments give the same partition (every rank computes it for itself)
        again = partition(pairs, rows, world)
        assert all(np.array_equal(x, y) for x, y in zip(shards, again))


def test_partition_edge_cases():
    pairs = exhaustive_pairs(range(3))
    shards = partition(pairs, None, 8)                      # more shards than pairs
    assert sorted(np.concatenate(shards).tolist()) == [0, 1, 2] and len(shards) == 8
    assert [len(s) for s in partition(np.zeros((0, 2), np.int64), {}, 4)] == [0, 0, 0, 0]
    one = partition(pairs, {0: 10, 1: 0, 2: 7}, 1)          # one shard: the list as it is, empty image and all
    assert one[0].tolist() == [0, 1, 2]
    # an arbitrary pair list (not exhaustive, not sorted)
    odd = np.array([[7, 2], [2, 9], [9, 7], [2, 7]], np.int64)
    shards = partition(odd, {2: 5, 7: 6, 9: 7}, 2)
    assert sorted(np.concatenate(shards).tolist()) == [0, 1, 2, 3]
    import pytest
    from sfm_amd import SfmHipError
    with pytest.raises(SfmHipError):
        partition(odd, {2: 5, 7: 6}, 2)                      # image 9 has no row count


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from sfm_amd import synth
from sfm_amd.distributed import exhaustive_pairs, gather_flat, images_of, match_pairs_sharded, partition
from oracle import oracle

class OracleMatcher:            # stands in for the GPU context: same call, CPU arithmetic
    def __init__(self, images): self.images = images
    def match_pairs(self, pairs, ratio, cross):
        return [oracle.match_pair(self.images[int(a)], self.images[int(b)], ratio, cross) for a, b in pairs]

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
sizes = (160, 100, 129, 64, 160, 0, 31)
images = {10 + i: im[:n] for i, (im, n) in enumerate(zip(synth.make_image_set(7, 160, 0.5, seed=21), sizes))}
rows = {i: len(im) for i, im in images.items()}
pairs = exhaustive_pairs(images.keys())
shards = partition(pairs, rows, world)
# this rank's matcher holds only the images of its shard
mine = {i: images[i] for i in images_of(pairs, shards[rank])}
got = match_pairs_sharded(OracleMatcher(mine), pairs, 0.8, True, rank, world, image_rows=rows)
if rank == 0:
    exp = OracleMatcher(images).match_pairs(pairs, 0.8, True)
    assert len(got) == len(exp) == 21
    for g, e in zip(got, exp):
        assert all(np.array_equal(x, y) for x, y in zip(g, e))
        assert g[2].dtype == np.float32 and g[2].view(np.uint32).tolist() == e[2].view(np.uint32).tolist()
    print("SHARDING_OK", sum(len(g[0]) for g in got))
else:
    assert got is None
# fewer pairs than ranks: a rank with an empty shard still takes part in the gather
few = exhaustive_pairs([10, 11])
got1 = match_pairs_sharded(OracleMatcher(images), few, 0.8, True, rank, world, image_rows=rows)
if rank == 0:
    assert len(got1) == 1 and all(np.array_equal(x, y) for x, y in zip(got1[0], exp[0]))
# the gather on its own, payload forced through several chunks, one rank with nothing to send
fake = [np.arange(0, 9, 2), np.arange(1, 9, 2)] if world == 2 else None
cnt = np.array([3 * q + 1 for q in fake[rank]], np.int32) if rank == 0 else np.zeros(4, np.int32)
vals = np.concatenate([np.full(c, 100 * q + 7, np.int32) for q, c in zip(fake[rank], cnt)] + [np.zeros(0, np.int32)])
back = gather_flat(np.stack([fake[rank].astype(np.int32), cnt], axis=1), [vals, vals.astype(np.float32)], fake, rank, world,
                   chunk_elems=5)
if rank == 0:
    head, off, (a, b) = back
    assert head[:, 0].tolist() == list(range(9))
    for q in range(9):
        want = [100 * q + 7] * (3 * q + 1) if q %% 2 == 0 else []
        assert a[off[q]:off[q + 1]].tolist() == want and b.view(np.float32)[off[q]:off[q + 1]].tolist() == want
    print("GATHER_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SHARDING_OK" in outs[0] and "GATHER_OK" in outs[0]


REPLICATE_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from sfm_amd import synth
from sfm_amd.distributed import replicate_images, shard_bounds

class Recorder:                 # stands in for the GPU context: remembers what was uploaded
    def __init__(self): self.images = {}
    def upload(self, image_id, rows): self.images[int(image_id)] = np.array(rows, dtype=np.float32, copy=True)

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
sizes = (160, 0, 129, 64, 1)
full = {10 + i: im[:n] for i, (im, n) in enumerate(zip(synth.make_image_set(5, 160, 0.5, seed=22), sizes))}
ids = sorted(full)
lo, hi = shard_bounds(len(ids), rank, world)
mine = {i: full[i] for i in ids[lo:hi]}          # each rank holds only its share
rec = Recorder()
rows = replicate_images(rec, mine, rank, world, method="collective")
assert rows == sum(sizes)
assert sorted(rec.images) == ids
for i in ids:
    assert rec.images[i].shape == full[i].shape and np.array_equal(rec.images[i], full[i]), i
host = Recorder()
assert replicate_images(host, full, rank, world, method="host") == sum(sizes)
assert all(np.array_equal(host.images[i], full[i]) for i in ids)
print("REPLICATE_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_replication_collective_two_ranks_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(REPLICATE_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all("REPLICATE_OK" in o for o in outs)


CASCADE_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from sfm_amd import synth
from sfm_amd.distributed import exhaustive_pairs, cascade_match_pairs_sharded
from oracle import cascade

class OracleCascade:            # stands in for the GPU context: same calls, CPU arithmetic
    def __init__(self, images): self.images, self.mean = images, None
    def cascade_prepare(self, pairs):
        self.mean = cascade.mean([self.images[int(i)] for i in cascade.image_order(pairs)])
    def cascade_match_pairs(self, pairs, cross):
        return [cascade.match_pair(self.images[int(a)], self.images[int(b)], self.mean, cross) for a, b in pairs]

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
imgs = synth.make_repetitive_set(4, 300, 0.6, 2, seed=23, sigma=0.002)
images = {10 + i: im[:n] for i, (im, n) in enumerate(zip(imgs, (300, 200, 0, 300)))}
pairs = exhaustive_pairs(images.keys())
got = cascade_match_pairs_sharded(OracleCascade(images), pairs, True, rank, world)
if rank == 0:
    whole, _, _ = cascade.match_set(images, pairs, True)     # the job on one process
    assert len(got) == len(pairs) == 6
    for (a, b), (nd, l, r, d) in zip(pairs.tolist(), got):
        wnd, wl, wr = whole[(a, b)]
        assert (nd, l.tolist(), r.tolist()) == (wnd, wl.tolist(), wr.tolist())
    print("CASCADE_SHARDING_OK", sum(len(g[1]) for g in got))
else:
    assert got is None
dist.barrier()
dist.destroy_process_group()
'''


def test_cascade_two_ranks_over_gloo(tmp_path):
    """The mean vector comes from the whole job's images on every rank, whatever slice it matches."""
    script = tmp_path / "worker.py"
    script.write_text(CASCADE_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29535", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "CASCADE_SHARDING_OK" in outs[0]
