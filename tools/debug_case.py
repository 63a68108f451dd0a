import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sfm_amd
from sfm_amd import synth
from oracle import oracle
m = sfm_amd.Matcher(0)
A, B, _ = synth.make_pair(300, 300, 150, seed=77)
B[10] = B[20]; B[200:210] = A[5]; A[100] = A[101]
for ratio, cross in ((0.8, True), (0.8, False), (5.0, False)):
    m.upload(0, A); m.upload(1, B)
    (l, r, d), = m.match_pairs([(0, 1)], ratio, cross)
    el, er, ed = oracle.match_pair(A, B, ratio, cross)
    print("ratio", ratio, "cross", cross, "n", len(l), len(el), "stats", m.stats()["rescanned_queries"])
    if l.tolist() != el.tolist():
        print(" lhs differ", set(l.tolist()) ^ set(el.tolist()))
    for i, (a, b) in enumerate(zip(r.tolist(), er.tolist())):
        if a != b:
            q = int(el[i])
            da = oracle.l2_metric_sq(A[q], B[a]); db = oracle.l2_metric_sq(A[q], B[b])
            print(" query", q, "got", a, "exp", b, "dsq got", da, "dsq exp", db, "gpu dist", d[i], "exp dist", ed[i])
            allsq = np.array([oracle.l2_metric_sq(A[q], B[j]) for j in range(300)])
            order = np.argsort(allsq, kind="stable")[:8]
            print("   top8", order.tolist(), allsq[order].tolist())
