"""Host cost of one sfm_hip_match_pairs call and the latency of a lone step (queue -> results)."""
import sys, time, ctypes, numpy as np
sys.path.insert(0, '.')
import sfm_amd
from sfm_amd import synth
m = sfm_amd.Matcher(0)
A, B, _ = synth.make_pair(8192, 8192, 4096, seed=2024)
m.upload(0, A); m.upload(1, B)
pairs = np.array([[0, 1]], np.int64)
for _ in range(300): m.match_pairs_async(pairs, 0.8, True, stride=8192)
m.sync()
host, lone = [], []
for _ in range(20):
    for _ in range(50): m.match_pairs_async(pairs, 0.8, True, stride=8192)
    m.sync()
    t0 = time.perf_counter()
    m.match_pairs_async(pairs, 0.8, True, stride=8192)
    t1 = time.perf_counter()
    m.sync()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e6); lone.append((t2 - t0) * 1e6)
print("host time of one call (python + C): med %.1f us" % np.median(host))
print("lone step, call to synced results: med %.1f us min %.1f" % (np.median(lone), min(lone)))
# the C call alone
lib = m._lib
cnt, lhs, rhs, dist = m._outputs(1, 8192)
args = (m._h, pairs.ctypes.data, 1, ctypes.c_double(0.8), 1, cnt.ptr, lhs.ptr, rhs.ptr, dist.ptr, 8192)
c = []
for _ in range(200):
    t0 = time.perf_counter(); lib.sfm_hip_match_pairs(*args); c.append((time.perf_counter() - t0) * 1e6)
m.sync()
print("ctypes call alone: med %.1f us" % np.median(c))
t0 = time.perf_counter()
for _ in range(20): m.match_pairs_async(pairs, 0.8, True, stride=8192)
m.sync()
print("20 steps after an idle sync: %.1f us per step" % ((time.perf_counter() - t0) / 20 * 1e6))
