"""Where the cascade-hash path's per-image preparation goes: column sums, hashing + grouping, for a fresh
set of images and for a set that is hashed again.   python tools/cascade_prepare_times.py [--images 32]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfm_amd  # noqa: E402
from sfm_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=32)
ap.add_argument("--features", type=int, default=8192)
a = ap.parse_args()
m = sfm_amd.Matcher(0)
m.cascade_set_projection(*sfm_amd.binding.cascade_reference_projection())
imgs = synth.make_repetitive_set(a.images, a.features, 0.5, 2, seed=7, sigma=0.002)
ids = list(range(a.images))
pairs = [(x, y) for x in ids for y in ids[x + 1:]]


def timed(f):
    t0 = time.perf_counter()
    r = f()
    return (time.perf_counter() - t0) * 1e3, r


for rep in range(3):
    m.drop_all()
    for i, d in zip(ids, imgs):
        m.upload(i, d)
    m.sync()
    t_sum, sums = timed(lambda: m.cascade_column_sums(ids))
    mean = np.zeros(128, np.float32)
    t_hash, _ = timed(lambda: m.cascade_hash_images(ids, mean))
    t_again, _ = timed(lambda: m.cascade_hash_images(ids, mean))
    t_prep, _ = timed(lambda: m.cascade_prepare(pairs))
    print("fresh images: column sums %.2f ms, hashing %.2f ms; hashing again %.2f ms; sfm_hip_cascade_prepare (again) %.2f ms  [%d images x %d]"
          % (t_sum, t_hash, t_again, t_prep, a.images, a.features), flush=True)
m.close()
