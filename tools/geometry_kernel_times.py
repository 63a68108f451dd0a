"""Per-kernel times of the geometric verification on the bench's two cases (library statistics with
profiling on): python tools/geometry_kernel_times.py [--matches 1000]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfm_amd  # noqa: E402
from sfm_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--matches", type=int, default=1000)
ap.add_argument("--pairs", type=int, default=496)
a = ap.parse_args()
m = sfm_amd.Matcher(0)
for frac in (0.7, 0.4):
    sets = [synth.make_two_view_matches(a.matches, frac, seed=500 + k, noise=0.3)[:2] for k in range(a.pairs)]
    offsets = np.zeros(a.pairs + 1, np.int64)
    offsets[1:] = np.cumsum([len(l) for l, _ in sets])
    lhs = np.concatenate([l for l, _ in sets])
    rhs = np.concatenate([r for _, r in sets])
    m.geometry_verify_flat(offsets, lhs, rhs, seed=1)
    t0 = time.perf_counter()
    for _ in range(3):
        m.geometry_verify_flat(offsets, lhs, rhs, seed=1)
    dt = (time.perf_counter() - t0) / 3
    m.set_profiling(True)
    m.geometry_reset_stats()
    F, flags, counts, rounds = m.geometry_verify_flat(offsets, lhs, rhs, seed=1)
    st = m.geometry_stats()
    m.set_profiling(False)
    print("%.0f %% inliers: call %.2f ms = %.0f pairs/s; fit %.2f ms, count %.2f ms; rounds mean %.0f; fitted rounds %d"
          % (100 * frac, dt * 1e3, a.pairs / dt, st["fit_ms"], st["count_ms"], float(np.mean(rounds)), st["rounds"]))
m.close()
