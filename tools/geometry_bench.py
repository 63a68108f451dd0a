"""Throughput of the geometric verification on one GPU, the CPU oracle timed beside it on a sample.

    python tools/geometry_bench.py [--pairs 496] [--matches 1000] [--inliers 0.7] [--cpu-pairs 8]

Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import sfm_amd  # noqa: E402
from sfm_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=496)
    ap.add_argument("--matches", type=int, default=1000)
    ap.add_argument("--inliers", type=float, default=0.7)
    ap.add_argument("--cpu-pairs", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    sets = [synth.make_two_view_matches(a.matches, a.inliers, seed=500 + k, noise=0.3)[:2] for k in range(a.pairs)]
    m = sfm_amd.Matcher(0)
    out = m.geometry_verify_pairs(sets, seed=1)      # warm-up, and the per-pair results for the checks below
    offsets = np.zeros(a.pairs + 1, np.int64)
    offsets[1:] = np.cumsum([len(l) for l, _ in sets])
    lhs = np.concatenate([l for l, _ in sets])
    rhs = np.concatenate([r for _, r in sets])
    t0 = time.perf_counter()
    for _ in range(a.reps):
        m.geometry_verify_flat(offsets, lhs, rhs, seed=1)      # one C-ABI call: upload, rounds, download
    dt = (time.perf_counter() - t0) / a.reps
    rounds = [o[3] for o in out]
    res = {"workload": "%d pairs x %d matches, %.0f %% planted inliers" % (a.pairs, a.matches, 100 * a.inliers),
           "pairs_per_s": a.pairs / dt, "seconds_per_call": dt, "accepted_pairs": sum(o[0] for o in out),
           "ransac_rounds_mean": float(np.mean(rounds)), "ransac_rounds_max": int(max(rounds)),
           "inliers_mean": float(np.mean([o[2].sum() for o in out]))}
    if a.cpu_pairs > 0:
        from oracle import geometry  # the checker, timed as the CPU baseline
        t0 = time.perf_counter()
        same = True
        for k in range(min(a.cpu_pairs, a.pairs)):
            ok, F, flags, r = geometry.compute_fundamental(sets[k][0], sets[k][1], seed=1, pair=k)
            same = same and ok == out[k][0] and r == out[k][3] and flags.tolist() == out[k][2].tolist() and \
                F.view(np.uint64).tolist() == out[k][1].view(np.uint64).tolist()
        dc = time.perf_counter() - t0
        res["cpu"] = {"pairs": min(a.cpu_pairs, a.pairs), "threads": 1, "pairs_per_s": min(a.cpu_pairs, a.pairs) / dc,
                      "equal_to_gpu": bool(same)}
    m.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
