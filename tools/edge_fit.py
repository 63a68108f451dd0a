"""Edge cost of a short timed region: K single-pair steps (8192 x 8192) between two synchronisations,
for several K, median of many repetitions; elapsed(K) = edge + K * step.
    python tools/edge_fit.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sfm_amd  # noqa: E402
from sfm_amd import synth  # noqa: E402

m = sfm_amd.Matcher(0)
A, B, _ = synth.make_pair(8192, 8192, 4096, seed=1)
m.upload(0, A)
m.upload(1, B)
pairs = np.array([[0, 1]], np.int64)


def region(k):
    m.sync()
    t0 = time.perf_counter()
    for _ in range(k):
        m.match_pairs_async(pairs, 0.8, True, stride=8192)
    t1 = time.perf_counter()
    m.sync()
    return (time.perf_counter() - t0) * 1e6, (t1 - t0) * 1e6


t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    region(20)
ks = [1, 2, 3, 4, 6, 8, 12, 20, 40]
res = {}
for k in ks:
    xs = [region(k) for _ in range(60)]
    res[k] = (float(np.median([x[0] for x in xs])), float(np.median([x[1] for x in xs])))
    print("K %3d  elapsed %8.1f us  (host calls returned after %7.1f us)  per step %6.1f" % (k, res[k][0], res[k][1], res[k][0] / k))
b = (res[40][0] - res[20][0]) / 20
print("step (K 20 -> 40): %.2f us;  edge at K = 20: %.1f us, at K = 1: %.1f us" % (b, res[20][0] - 20 * b, res[1][0] - b))

# what the bench's region has around the same 20 steps
import torch  # noqa: E402

torch.cuda.synchronize()


def variant(name, sync_before=False, sync_after=False, profiling=False, idle_us=0.0, reset=False):
    xs = []
    for _ in range(60):
        m.set_profiling(profiling)
        m.sync()
        if reset:
            m.reset_stats()
        if sync_before:
            torch.cuda.synchronize()
        if idle_us:
            t = time.perf_counter()
            while (time.perf_counter() - t) * 1e6 < idle_us:
                pass
        t0 = time.perf_counter()
        for _ in range(20):
            m.match_pairs_async(pairs, 0.8, True, stride=8192)
        m.sync()
        if sync_after:
            torch.cuda.synchronize()
        xs.append((time.perf_counter() - t0) * 1e6)
    m.set_profiling(False)
    print("%-58s %8.1f us" % (name, float(np.median(xs))))


variant("20 steps, library sync only")
variant("+ torch.cuda.synchronize() after", sync_after=True)
variant("+ torch.cuda.synchronize() before", sync_before=True)
variant("+ both", sync_before=True, sync_after=True)
variant("+ kernel timing on", profiling=True)
variant("+ 150 us of idle GPU before", idle_us=150.0)
variant("+ 1000 us of idle GPU before", idle_us=1000.0)
variant("all of the bench's (timing, reset, both, 150 us)", sync_before=True, sync_after=True, profiling=True, idle_us=150.0, reset=True)
m.close()
