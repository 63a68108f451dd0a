"""Randomised parity run of the HIP path against the CPU oracle, beyond what the test suite holds:
    python tools/parity_stress.py [--seconds 90] [--seed 1]
Random image sizes and four kinds of descriptors (RootSIFT-like rows with planted near-copies, Gaussian
rows, rows on a small integer lattice -- full of exact ties --, crowds of near-duplicates), both distance
engines, with and without cross-check, ratios near and far from 0.8.  Indices and distance bits must be
the oracle's.  Prints one line per hundred pairs and a summary; exits non-zero on the first difference."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import sfm_amd  # noqa: E402
from sfm_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402  (the checker)


def make(kind, na, nb, rng):
    if kind == 0:
        A, B, _ = synth.make_pair(na, nb, min(na, nb) // 2, seed=int(rng.integers(1 << 30)))
        return A, B
    if kind == 1:
        return rng.standard_normal((na, 128)).astype(np.float32), rng.standard_normal((nb, 128)).astype(np.float32)
    if kind == 2:   # lattice: coordinates in {0, 1, 2}, many equal distances
        return rng.integers(0, 3, (na, 128)).astype(np.float32), rng.integers(0, 3, (nb, 128)).astype(np.float32)
    base = synth.rootsift_rows(max(4, min(na, nb) // 6), int(rng.integers(1 << 30)))
    pick = lambda n: (base[rng.integers(0, len(base), n)] + rng.standard_normal((n, 128)) * 10.0 ** rng.uniform(-6, -2)).astype(np.float32)
    return pick(na), pick(nb)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=90.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    m = sfm_amd.Matcher(0)
    t0 = time.perf_counter()
    n = 0
    rescans = 0
    while time.perf_counter() - t0 < a.seconds:
        na, nb = (int(2 ** rng.uniform(0, 11.7)) for _ in range(2))
        kind = int(rng.integers(0, 4))
        A, B = make(kind, na, nb, rng)
        ratio = float(rng.choice([0.8, 0.8, 0.8, 0.6, 0.95, 1.0]))
        cross = bool(rng.integers(0, 2))
        want = oracle.match_pair(A, B, ratio, cross, threads=8)
        m.upload(0, A)
        m.upload(1, B)
        for engine in ("f32", "f16x3"):
            m.set_engine(engine)
            m.reset_stats()
            (l, r, d), = m.match_pairs([(0, 1)], ratio, cross)
            rescans += m.stats()["rescanned_queries"]
            if l.tolist() != want[0].tolist() or r.tolist() != want[1].tolist() or \
                    d.view(np.uint32).tolist() != want[2].view(np.uint32).tolist():
                print("DIFFERENCE: kind %d, %d x %d, ratio %g, cross %s, engine %s, seed %d, pair %d"
                      % (kind, na, nb, ratio, cross, engine, a.seed, n))
                print("got ", l.tolist(), r.tolist(), d.view(np.uint32).tolist())
                print("want", want[0].tolist(), want[1].tolist(), want[2].view(np.uint32).tolist())
                out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
                os.makedirs(out, exist_ok=True)
                np.savez(os.path.join(out, "parity_difference.npz"), A=A, B=B, ratio=ratio, cross=cross)
                sys.exit(1)
        n += 1
        if n % 100 == 0:
            print("%d pairs identical on both engines (%.0f s)" % (n, time.perf_counter() - t0), flush=True)
    m.close()
    print("done: %d pairs, both engines, identical to the oracle; %d queries took the exact scan" % (n, rescans))


if __name__ == "__main__":
    main()
