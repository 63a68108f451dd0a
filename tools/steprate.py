"""Steady-state time per single-pair step (300 steps between two syncs, after 400 warm-up steps)."""
import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import sfm_amd
from sfm_amd import synth
m = sfm_amd.Matcher(0)
if len(sys.argv) > 1: m.set_engine(sys.argv[1])
A, B, truth = synth.make_pair(8192, 8192, 4096, seed=2024)
m.upload(0, A); m.upload(1, B)
pairs = np.array([[0, 1]], np.int64)
for _ in range(400): m.match_pairs_async(pairs, 0.8, True, stride=8192)
m.sync()
res = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(300): m.match_pairs_async(pairs, 0.8, True, stride=8192)
    m.sync()
    res.append((time.perf_counter() - t0) / 300 * 1e6)
print("us per step:", " ".join("%.1f" % v for v in res))
