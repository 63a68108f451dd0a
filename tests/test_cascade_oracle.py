"""CPU-only checks of the cascade-hash oracle (oracle/cascade_oracle.cc).

The reference holds no test, fixture or recorded output for CascadeHashMatcher and cannot be built
here (Eigen is absent), so parity with the reference is unpinned; what can be done is done here:
  * a known-answer case derived by hand from the reference's text (tests/cascade_kat.py);
  * the random projections re-derived by an independent restatement of the two standard-library
    components the reference uses (MT19937 and libstdc++'s polar normal_distribution);
  * the whole flow cross-checked against a second, independent numpy restatement on small inputs;
  * the library's host-only helpers (projection, image order) equal to the oracle's.
"""
import math

import numpy as np
import pytest

from cascade_kat import build as build_kat, coordinate_projection
from oracle import cascade
from sfm_amd import synth


@pytest.fixture
def coordinate_hashes():
    p, s = coordinate_projection()
    cascade.set_projection(p, s)
    yield
    cascade.set_projection(None, None)


def test_known_answer_case_on_the_oracle(coordinate_hashes):
    A, B, expected = build_kat()
    zero = np.zeros(128, np.float32)
    codes, bids = cascade.hash_image(A, zero)
    # coordinate projections: hash bit i = sign of dimension i; first row of a group = top bit
    assert all(((int(codes[r, i // 32]) >> (i % 32)) & 1) == int(A[r, i] > 0) for r in range(len(A)) for i in range(128))
    assert all(int(bids[r, g]) == sum(int(A[r, 10 * g + k] > 0) << (9 - k) for k in range(10))
               for r in range(len(A)) for g in range(6))
    for cross in (False, True):
        nd, l, r, d = cascade.match_pair(A, B, zero, cross)
        assert (nd, list(zip(l.tolist(), r.tolist()))) == expected[cross]
    nd, l, r, d = cascade.match_pair(A, B, zero, False)
    # L2^2 = (v_q - v_t)^2 + flips * 2^-18 (tests/cascade_kat.py), e.g. A2 -> B2: 0.25^2 + 3 * 2^-18
    assert d[0] == np.float32(math.sqrt(0.0625 + 3 * 2.0 ** -18))
    assert d[2] == np.float32(1.0)      # A4 -> B3


# ---- the projection matrices, from an independent restatement of the generator ------------------

class _MT19937:
    """32-bit Mersenne Twister as std::mt19937 specifies it (seed 5489 = std::mt19937::default_seed)."""

    def __init__(self, seed=5489):
        self.mt = [0] * 624
        self.mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.idx = 624

    def __call__(self):
        if self.idx >= 624:
            mt = self.mt
            for k in range(624):
                y = (mt[k] & 0x80000000) | (mt[(k + 1) % 624] & 0x7FFFFFFF)
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.idx = 0
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y


def _libstdcxx_normals(count):
    """std::normal_distribution<double>(0, 1) of libstdc++ over std::mt19937: Marsaglia's polar
    method on two std::generate_canonical<double, 53> draws (two 32-bit words each, low word
    first), returning y * mult first and keeping x * mult for the next call."""
    gen = _MT19937()

    def canonical():
        lo, hi = gen(), gen()
        return (lo + hi * 4294967296.0) / 18446744073709551616.0

    out, saved = [], None
    while len(out) < count:
        if saved is not None:
            out.append(saved)
            saved = None
            continue
        while True:
            x = 2.0 * canonical() - 1.0
            y = 2.0 * canonical() - 1.0
            r2 = x * x + y * y
            if 0.0 < r2 <= 1.0:
                break
        mult = math.sqrt(-2.0 * math.log(r2) / r2)
        saved = x * mult
        out.append(y * mult)
    return np.array(out, np.float64)


def test_projection_equals_an_independent_restatement_of_the_generator():
    p, s = cascade.projection()
    draws = _libstdcxx_normals(128 * 128 + 6 * 10 * 128).astype(np.float32)
    assert draws[:128 * 128].view(np.uint32).tolist() == p.reshape(-1).view(np.uint32).tolist()
    assert draws[128 * 128:].view(np.uint32).tolist() == s.reshape(-1).view(np.uint32).tolist()


def test_library_helpers_equal_the_oracle():
    from sfm_amd import binding
    p, q = binding.cascade_reference_projection()
    op, oq = cascade.projection()
    assert p.view(np.uint32).tolist() == op.view(np.uint32).tolist()
    assert q.view(np.uint32).tolist() == oq.view(np.uint32).tolist()
    for pairs in ([(0, 1)], [(0, 1), (0, 2), (1, 2), (5, 9), (3, 1000000007), (40, 41)],
                  [(a, b) for a in range(40) for b in range(a + 1, 40)]):
        assert binding.cascade_image_order(pairs).tolist() == cascade.image_order(pairs).tolist()
    # the order is a property of the container, not of the pair order handed in
    assert sorted(cascade.image_order([(7, 9), (1, 2)]).tolist()) == [1, 2, 7, 9]


# ---- the whole flow against a second restatement -------------------------------------------------

def _seq_sum_f32(terms):
    """Sum over the last axis in index order, one float32 accumulator per row."""
    acc = np.zeros(terms.shape[:-1], np.float32)
    for j in range(terms.shape[-1]):
        acc = (acc + terms[..., j]).astype(np.float32)
    return acc


def _second_restatement(images, pairs, cross):
    """numpy/dict restatement of CascadeHashMatcher::Match (cascade_hash_matcher.cc:118-205),
    written from the reference's text independently of the C++ oracle."""
    P, S = cascade.projection()
    order = cascade.image_order(pairs)      # container order is library-defined: taken from the oracle
    acc = np.zeros(128, np.float32)
    total = 0
    for i in order:
        raw = images[int(i)].reshape(-1)
        n = images[int(i)].shape[0]
        # column-major view of the row-major block: column c = c-th run of n floats (.cc:160-162)
        cols = _seq_sum_f32(raw.reshape(128, n)) if n else np.zeros(128, np.float32)
        acc = (acc + cols).astype(np.float32)
        total += n
    mean = (acc / np.float32(total)).astype(np.float32)

    def hashes(x):
        d = (x - mean).astype(np.float32)
        prim = np.stack([_seq_sum_f32((P[k][None, :] * d).astype(np.float32)) > 0 for k in range(128)], axis=1)
        bid = np.zeros((len(x), 6), np.int64)
        for g in range(6):
            for k in range(10):
                bit = _seq_sum_f32((S[g, k][None, :] * d).astype(np.float32)) > 0
                bid[:, g] = (bid[:, g] << 1) + bit
        return prim, bid

    def l2(a, b):
        s = np.float32(0)
        for k in range(128):
            t = np.float32(a[k] - b[k])
            s = np.float32(s + np.float32(t * t))
        return np.float32(np.sqrt(s))

    def one_way(xa, ha, xb, hb):
        (ca, ba), (cb, bb) = ha, hb
        out = []
        for i in range(len(xa)):
            pool = [j for g in range(6) for j in range(len(xb)) if bb[j, g] == ba[i, g]]
            if len(pool) < 2:
                continue
            bins = {}
            for j in dict.fromkeys(pool):
                bins.setdefault(int((ca[i] != cb[j]).sum()), []).append(j)
            short = []
            for h in range(11):
                if len(short) >= 10:
                    break
                short += [(l2(xa[i], xb[j]), j) for j in bins.get(h, [])]
            if len(short) >= 2:
                out.append((i, min(short)[1]))
        return out

    H = {i: hashes(images[i]) for i in images}
    result = {}
    for a, b in pairs:
        fwd = one_way(images[a], H[a], images[b], H[b])
        if cross:
            back = dict(one_way(images[b], H[b], images[a], H[a]))
            result[(a, b)] = (len(fwd), [(i, j) for i, j in fwd if back.get(j) == i])
        else:
            result[(a, b)] = (0, fwd)
    return mean, result


@pytest.mark.parametrize("cross", [True, False])
def test_whole_flow_equals_a_second_restatement(cross):
    imgs = synth.make_repetitive_set(3, 150, 0.7, 3, seed=9, sigma=0.002)
    images = {4: imgs[0], 9: imgs[1][:120], 11: imgs[2][:77]}
    pairs = [(4, 9), (4, 11), (9, 11)]
    mean, want = _second_restatement(images, pairs, cross)
    got_mean = cascade.mean([images[int(i)] for i in cascade.image_order(pairs)])
    assert got_mean.view(np.uint32).tolist() == mean.view(np.uint32).tolist()
    whole, _, _ = cascade.match_set(images, pairs, cross)
    n = 0
    for pr in pairs:
        nd, l, r = whole[pr]
        assert (nd, list(zip(l.tolist(), r.tolist()))) == want[pr]
        one = cascade.match_pair(images[pr[0]], images[pr[1]], mean, cross)
        assert (one[0], one[1].tolist(), one[2].tolist()) == (nd, l.tolist(), r.tolist())
        n += len(l)
    assert n > 20
    # threads only change who does the work
    whole4, _, _ = cascade.match_set(images, pairs, cross, threads=4)
    assert all(whole4[p][0] == whole[p][0] and whole4[p][1].tolist() == whole[p][1].tolist() for p in pairs)


def test_mean_uses_the_reference_view_of_the_block():
    """cascade_hash_matcher.cc:160: the row-major block is mapped as a column-major matrix, so the
    "column sums" are sums of runs of n consecutive floats, not per-dimension sums."""
    x = np.zeros((4, 128), np.float32)
    x[0, :] = 1.0          # the first row fills the first 128 floats of the block
    sums = cascade.image_column_sums(x)
    # runs have length 4: the first 32 runs cover row 0 (sum 4 each), the rest are zero
    assert sums.tolist() == [4.0] * 32 + [0.0] * 96


def test_golden_cases_reproduce():
    """tests/golden/cascade_cases.json: the pair (generator, oracle) gives the same bits here as where
    the file was written (tests/golden/make_golden_next.py)."""
    import json
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden_next as g
    with open(os.path.join(ROOT, "tests", "golden", "cascade_cases.json")) as f:
        want = json.load(f)
    p, s = cascade.projection()
    assert g.digest(p.view(np.uint32), s.view(np.uint32)) == want["projection_bits_sha256"]
    for c, w in zip(g.CASCADE_CASES, want["cases"]):
        assert g.cascade_case(c) == w, c[0]
    assert sum(pr["count"] for w in want["cases"] for pr in w["pairs"]) > 300


def test_measures_equal_the_reference_build_bit_for_bit():
    """The two measures the cascade matcher uses -- L2_metric and HammingDistance, both in the
    reference's src/metric.hpp -- against that header compiled from the reference checkout
    (oracle/_ref, built by oracle/Makefile).  This is the part of this matcher that IS pinned."""
    import ctypes
    from oracle import oracle
    ref = oracle.ref_metric_lib()
    if ref is None or not hasattr(ref, "ref_HammingDistance128"):
        pytest.skip("oracle/_ref was not built (no reference checkout on this machine)")
    lib = cascade.lib()
    rng = np.random.default_rng(11)
    base = rng.integers(0, 2 ** 32, size=(300, 4), dtype=np.uint64).astype(np.uint32)
    other = base.copy()
    for r in range(300):           # from identical to unrelated
        flips = rng.integers(0, 128, size=r % 40)
        for f in flips:
            other[r, f // 32] ^= np.uint32(1 << (f % 32))
    other[200:] = rng.integers(0, 2 ** 32, size=(100, 4), dtype=np.uint64).astype(np.uint32)
    for a, b in zip(base, other):
        want = ref.ref_HammingDistance128(a.ctypes.data, b.ctypes.data)
        assert lib.ch_hamming(a.ctypes.data, b.ctypes.data) == want
        assert want == sum(bin(int(x) ^ int(y)).count("1") for x, y in zip(a, b))
    rows = synth.rootsift_rows(200, 5)
    fp = ctypes.POINTER(ctypes.c_float)
    for k in range(100):
        a, b = rows[k], rows[199 - k]
        want = np.float32(ref.ref_L2_metric(a.ctypes.data_as(fp), b.ctypes.data_as(fp)))
        got = np.float32(lib.ch_l2_metric(a.ctypes.data, b.ctypes.data))
        assert got.view(np.uint32) == want.view(np.uint32)
