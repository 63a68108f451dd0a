"""CPU-only: the multi-rank path (static pair sharding + gather) over gloo with two processes."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT
from sfm_amd.distributed import exhaustive_pairs, shard_bounds


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


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from sfm_amd import synth
from sfm_amd.distributed import exhaustive_pairs, match_pairs_sharded
from oracle import oracle

class OracleMatcher:            # stands in for the GPU context: same call, CPU arithmetic
    def __init__(self, images): self.images = images
    def match_pairs(self, pairs, ratio, cross):
        return [oracle.match_pair(self.images[int(a)], self.images[int(b)], ratio, cross) for a, b in pairs]

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
images = {10 + i: im[:n] for i, (im, n) in enumerate(zip(synth.make_image_set(5, 160, 0.5, seed=21), (160, 100, 129, 64, 160)))}
pairs = exhaustive_pairs(images.keys())
got = match_pairs_sharded(OracleMatcher(images), pairs, 0.8, True, rank, world)
if rank == 0:
    exp = OracleMatcher(images).match_pairs(pairs, 0.8, True)
    assert len(got) == len(exp) == 10
    for g, e in zip(got, exp):
        assert all(np.array_equal(x, y) for x, y in zip(g, e))
    print("SHARDING_OK", sum(len(g[0]) for g in got))
else:
    assert got is None
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
    assert "SHARDING_OK" in outs[0]
