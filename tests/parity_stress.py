"""Randomised parity run of the HIP path against the CPU oracle, beyond what the test suite holds:
    python tests/parity_stress.py [--path brute|batch|cascade|geometry] [--seconds 90] [--seed 1]
brute (the default): random image sizes and four kinds of descriptors (RootSIFT-like rows with planted near-copies, Gaussian
rows, rows on a small integer lattice -- full of exact ties --, crowds of near-duplicates), both distance
engines, with and without cross-check, ratios near and far from 0.8.  Indices and distance bits must be
the oracle's.  batch: sets of three to seven images of random sizes that share rows, every ordered pair in one
call of sfm_hip_match_pairs (the batched kernels), and one pair of the set through
sfm_hip_match_pair_noalloc.  cascade: sets of two to four images (repeated structure with noise of random size, plain
RootSIFT-like rows, crowds of near-duplicates, lattice rows; sizes 0 ... 3000), every pair of a set, with and
without cross-check: default records, indices and distance bits.  geometry: calls of one to six pairs of 0 ...
3000 matches with random inlier fractions and noise: verdict, matrix bits, flags, rounds.
Prints a line every so many pairs and a summary; exits non-zero on the first difference (the inputs go to
gpurun_out/parity_difference.npz)."""
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


def dump(**arrays):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    np.savez(os.path.join(out, "parity_difference.npz"), **arrays)


def cascade_image(kind, n, rng, shared):
    if kind == 0:
        return shared[int(rng.integers(len(shared)))][:n]
    if kind == 1:
        return synth.rootsift_rows(n, int(rng.integers(1 << 30)))
    if kind == 2:
        return rng.integers(0, 3, (n, 128)).astype(np.float32)
    base = synth.rootsift_rows(max(4, n // 6), int(rng.integers(1 << 30)))
    return (base[rng.integers(0, len(base), n)] + rng.standard_normal((n, 128)) * 10.0 ** rng.uniform(-6, -2)).astype(np.float32)


def run_cascade(a):
    from oracle import cascade
    rng = np.random.default_rng(a.seed)
    m = sfm_amd.Matcher(0)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < a.seconds:
        k = int(rng.integers(2, 5))
        kind = int(rng.integers(0, 4))
        shared = synth.make_repetitive_set(k, 3000, float(rng.uniform(0.2, 0.9)), int(rng.integers(1, 5)),
                                           seed=int(rng.integers(1 << 30)), sigma=float(10.0 ** rng.uniform(-4, -2)))
        images = {}
        for i in range(k):
            size = int(2 ** rng.uniform(0, 11.55)) if rng.integers(0, 12) else int(rng.integers(0, 3))
            images[3 * i + 1] = np.ascontiguousarray(cascade_image(kind, size, rng, shared))
        ids = sorted(images)
        pairs = [(x, y) for u, x in enumerate(ids) for y in ids[u + 1:]]
        m.drop_all()
        for i, d in images.items():
            m.upload(i, d)
        m.cascade_prepare(pairs)
        order = cascade.image_order(pairs)
        mean = cascade.mean([images[int(i)] for i in order])
        for cross in (False, True):
            got = m.cascade_match_pairs(pairs, cross_check=cross)
            for (x, y), (nd, l, r, d) in zip(pairs, got):
                wnd, wl, wr, wd = cascade.match_pair(images[x], images[y], mean, cross)
                if nd != wnd or l.tolist() != wl.tolist() or r.tolist() != wr.tolist() or \
                        d.view(np.uint32).tolist() != wd.view(np.uint32).tolist():
                    print("DIFFERENCE: cascade, kind %d, images %s, pair (%d, %d), cross %s, seed %d, set %d"
                          % (kind, [len(images[i]) for i in ids], x, y, cross, a.seed, n))
                    dump(mean=mean, cross=cross, pair=np.array([x, y]), **{"image%d" % i: images[i] for i in ids})
                    sys.exit(1)
        n += 1
        if n % 20 == 0:
            print("%d sets identical (%.0f s)" % (n, time.perf_counter() - t0), flush=True)
    st = m.cascade_stats()
    m.close()
    print("done: %d sets of images, every pair with and without cross-check, identical to the oracle; "
          "list overflows %d" % (n, st["list_overflows"]))


def run_batch(a):
    rng = np.random.default_rng(a.seed)
    m = sfm_amd.Matcher(0)
    cap = 8192
    bufs = [m.pinned((1,), np.int32), m.pinned((cap,), np.int32), m.pinned((cap,), np.int32), m.pinned((cap,), np.float32)]
    t0 = time.perf_counter()
    n = 0
    n_pairs = 0
    while time.perf_counter() - t0 < a.seconds:
        k = int(rng.integers(3, 8))
        kind = int(rng.integers(0, 4))
        pool = synth.rootsift_rows(2500, int(rng.integers(1 << 30)))
        images = {}
        for i in range(k):
            size = int(2 ** rng.uniform(0, 11.2)) if rng.integers(0, 12) else int(rng.integers(0, 3))
            if kind == 0:      # noisy copies of rows of a common pool, and rows of its own
                own = synth.rootsift_rows(size, int(rng.integers(1 << 30)))
                take = rng.random(size) < 0.5
                d = np.where(take[:, None], pool[rng.integers(0, len(pool), size)] + rng.standard_normal((size, 128)) * 0.01, own)
            elif kind == 1:
                d = rng.standard_normal((size, 128))
            elif kind == 2:
                d = rng.integers(0, 3, (size, 128))
            else:              # crowds of near-duplicates of a few pool rows
                d = pool[rng.integers(0, max(4, size // 6), size)] + rng.standard_normal((size, 128)) * 10.0 ** rng.uniform(-6, -2)
            images[5 * i + 2] = np.ascontiguousarray(d, dtype=np.float32)
        ids = sorted(images)
        pairs = [(x, y) for x in ids for y in ids if x != y]
        ratio = float(rng.choice([0.8, 0.8, 0.6, 0.95, 1.0]))
        cross = bool(rng.integers(0, 2))
        want = [oracle.match_pair(images[x], images[y], ratio, cross, threads=8) for x, y in pairs]
        m.drop_all()
        for i, d in images.items():
            m.upload_async(i, d)
        for engine in ("f32", "f16x3"):
            m.set_engine(engine)
            got = m.match_pairs(pairs, ratio, cross)
            pick = int(rng.integers(len(pairs)))
            m.match_pair_noalloc_async(pairs[pick][0], pairs[pick][1], ratio, cross, *bufs)
            m.sync()
            c = int(bufs[0].array[0])
            got.append((bufs[1].array[:c].astype(np.int64), bufs[2].array[:c].astype(np.int64), bufs[3].array[:c].copy()))
            for which, ((x, y), (l, r, d), w) in enumerate(zip(pairs + [pairs[pick]], got, want + [want[pick]])):
                if l.tolist() != w[0].tolist() or r.tolist() != w[1].tolist() or \
                        d.view(np.uint32).tolist() != w[2].view(np.uint32).tolist():
                    print("DIFFERENCE: batch, kind %d, images %s, pair (%d, %d)%s, ratio %g, cross %s, engine %s, seed %d, set %d"
                          % (kind, [len(images[i]) for i in ids], x, y, " through the allocation-free entry point" if which == len(pairs) else "",
                             ratio, cross, engine, a.seed, n))
                    dump(A=images[x], B=images[y], ratio=ratio, cross=cross, **{"image%d" % i: images[i] for i in ids})
                    sys.exit(1)
        n += 1
        n_pairs += len(pairs)
        if n % 20 == 0:
            print("%d sets, %d pairs identical on both engines (%.0f s)" % (n, n_pairs, time.perf_counter() - t0), flush=True)
    m.sync()
    for b in bufs:
        b.free()
    m.close()
    print("done: %d sets, %d pairs in batches and %d through the allocation-free entry point, both engines, identical to the oracle"
          % (n, n_pairs, n))


def run_geometry(a):
    from oracle import geometry
    rng = np.random.default_rng(a.seed)
    m = sfm_amd.Matcher(0)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < a.seconds:
        sets = []
        for _ in range(int(rng.integers(1, 7))):
            size = int(2 ** rng.uniform(0, 11.55)) if rng.integers(0, 10) else int(rng.integers(0, 9))
            noise = float(rng.choice([0.0, 0.1, 0.5, 2.0]))
            l, r, _ = synth.make_two_view_matches(size, float(rng.uniform(0.1, 1.0)), seed=int(rng.integers(1 << 30)), noise=noise)
            if rng.integers(0, 8) == 0 and size > 1:      # repeated matches: degenerate samples
                pick = rng.integers(0, max(1, size // 8), size)
                l, r = np.ascontiguousarray(l[pick]), np.ascontiguousarray(r[pick])
            sets.append((l, r))
        seed = int(rng.integers(1 << 30))
        first = int(rng.integers(0, 1000))
        got = m.geometry_verify_pairs(sets, seed=seed, first_pair_index=first)
        for k, ((l, r), (ok, F, flags, rounds)) in enumerate(zip(sets, got)):
            wok, wF, wflags, wrounds = geometry.compute_fundamental(l, r, seed=seed, pair=first + k)
            if ok != wok or rounds != wrounds or F.view(np.uint64).tolist() != wF.view(np.uint64).tolist() or \
                    flags.tolist() != wflags.tolist():
                print("DIFFERENCE: geometry, %d matches, seed %d, pair index %d, run seed %d, call %d: rounds %d / %d, verdict %s / %s"
                      % (len(l), seed, first + k, a.seed, n, rounds, wrounds, ok, wok))
                dump(l=l, r=r, seed=seed, pair=first + k)
                sys.exit(1)
            n += 1
        if n % 50 < len(sets):
            print("%d pairs identical (%.0f s)" % (n, time.perf_counter() - t0), flush=True)
    m.close()
    print("done: %d pairs, identical to the oracle" % n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--path", choices=("brute", "batch", "cascade", "geometry"), default="brute")
    ap.add_argument("--seconds", type=float, default=90.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-log2", type=float, default=11.7, help="brute: image sizes are 2^u, u uniform in [0, this]")
    a = ap.parse_args()
    {"brute": run_brute, "batch": run_batch, "cascade": run_cascade, "geometry": run_geometry}[a.path](a)


def run_brute(a):
    rng = np.random.default_rng(a.seed)
    m = sfm_amd.Matcher(0)
    t0 = time.perf_counter()
    n = 0
    rescans = 0
    while time.perf_counter() - t0 < a.seconds:
        na, nb = (int(2 ** rng.uniform(0, a.max_log2)) for _ in range(2))
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
                dump(A=A, B=B, ratio=ratio, cross=cross)
                sys.exit(1)
        n += 1
        if n % 100 == 0:
            print("%d pairs identical on both engines (%.0f s)" % (n, time.perf_counter() - t0), flush=True)
    m.close()
    print("done: %d pairs, both engines, identical to the oracle; %d queries took the exact scan" % (n, rescans))


if __name__ == "__main__":
    main()
