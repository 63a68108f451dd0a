"""Known-answer case for the cascade-hash matching rule, worked out by hand from the text of
src/feature_matcher/cascade_hash_matcher.cc:51-116 and :177-195 (the reference holds no test or
fixture for this matcher, and cannot be built here).

Hash bits and distances are steered independently by replacing the random projections with
coordinate projections: hash bit i is the sign of dimension i, and bit k of group g's bucket id is
the sign of dimension 10 g + k.  The mean vector is zero.  Every row has magnitude 2^-10 in
dimensions 0..126 and carries its "position" v > 0 in dimension 127, so that for a query q and a
target t

    Hamming(q, t) = number of sign flips,   L2(q, t)^2 = (v_q - v_t)^2 + flips * 2^-18,

and with positions at least 1/8 apart the order of distances is the order of |v_q - v_t|.

Case c uses the 10-bit pattern c + 1 (bits 0..4 only) in every group, so rows of different cases
never share a bucket; flips inside the bucket dimensions use bit 9 of a group, which cannot turn
one case's pattern into another's.
"""
import numpy as np

MAG = 2.0 ** -10


def coordinate_projection():
    primary = np.zeros((128, 128), np.float32)
    primary[np.arange(128), np.arange(128)] = 1.0
    secondary = np.zeros((6, 10, 128), np.float32)
    for g in range(6):
        for k in range(10):
            secondary[g, k, 10 * g + k] = 1.0
    return primary, secondary


def _row(case, v, flips=()):
    x = np.full(128, MAG, np.float32)
    pattern = case + 1
    for g in range(6):
        for k in range(10):
            # bucket id bit for row k of a group is the (9 - k)-th bit of the id: the first row is the
            # most significant bit (cascade_hash_matcher.cc:33-35); only equality matters here
            if not (pattern >> k) & 1:
                x[10 * g + k] = -MAG
    for f in flips:
        x[f] = -x[f]
    x[127] = v
    return x


def build():
    """Returns (A, B, expected) with expected = {cross_check: (n_default, [(lhs_idx, rhs_idx), ...])}."""
    A, B = [], []
    # case 0 -- A0: nothing of the second image in any of its buckets: no match (.cc:80-82)
    A.append(_row(0, 1.0))
    # case 1 -- A1: one target, found through all six groups.  The flat candidate list has six
    # entries (so :80 does not stop it) but one distinct candidate: the short list has one entry
    # and :105 needs two.  No match.
    A.append(_row(1, 1.0))
    B.append(_row(1, 1.5))                                   # B0
    # case 2 -- A2 and a decoy A3 in the same buckets; B1 at Hamming 0, B2 at Hamming 3.
    # Both targets enter the short list of either query (bins are added up to distance 10 while
    # the list is shorter than ten).  A2 (v 1): B1 at 1.5, B2 at 0.25 -> B2.  A3 (v 5): B1 at 2.5,
    # B2 at 3.75 -> B1.  No ratio test (:105-113).  Backwards: B1 sees A2 at 1.5, A3 at 2.5 -> A2;
    # B2 sees A2 at 0.25, A3 at 3.75 -> A2.
    A.append(_row(2, 1.0))                                   # A2
    A.append(_row(2, 5.0))                                   # A3
    B.append(_row(2, 2.5))                                   # B1
    B.append(_row(2, 1.25, flips=(60, 61, 62)))              # B2
    # case 3 -- A4 with two identical targets: equal distances, std::pair comparison picks the
    # lower index (:106-111).
    A.append(_row(3, 1.0))                                   # A4
    B.append(_row(3, 2.0))                                   # B3
    B.append(_row(3, 2.0))                                   # B4
    # case 4 -- A5: four targets at Hamming 0 (B5..B8), seven at Hamming 1 (B9..B15), and the
    # nearest of all at Hamming 2 (B16).  Bins 0 and 1 give eleven entries, the loop condition
    # `size < 10` then fails and bin 2 is never looked at (:95-97): the winner is the nearest of
    # the eleven, B9 at distance 1.
    A.append(_row(4, 1.0))                                   # A5
    for v in (3.0, 3.25, 3.5, 3.75):
        B.append(_row(4, v))                                 # B5..B8
    for k in range(7):
        B.append(_row(4, 2.0 + 0.25 * k, flips=(60,)))       # B9..B15
    B.append(_row(4, 1.125, flips=(60, 61)))                 # B16
    # case 5 -- A6: the nearest target (B17) is at Hamming 11, beyond the last bin the loop reads
    # (k <= bits_of_group = 10, :95); two more at Hamming 10 -> B18.
    A.append(_row(5, 1.0))                                   # A6
    B.append(_row(5, 1.125, flips=tuple(range(60, 71))))     # B17
    B.append(_row(5, 2.0, flips=tuple(range(60, 70))))       # B18
    B.append(_row(5, 3.0, flips=tuple(range(60, 70))))       # B19
    # case 6 -- A7: one target at Hamming 0, one at Hamming 11: the short list has one entry.  No match.
    A.append(_row(6, 1.0))                                   # A7
    B.append(_row(6, 2.0))                                   # B20
    B.append(_row(6, 1.5, flips=tuple(range(60, 71))))       # B21
    # case 7 -- A8: B22 shares only group 3's bucket (bit 9 flipped in groups 0, 1, 2, 4, 5:
    # Hamming 5), B23 shares all six.  Both are candidates; B22 is nearer.
    A.append(_row(7, 1.0))                                   # A8
    B.append(_row(7, 1.25, flips=(9, 19, 29, 49, 59)))       # B22
    B.append(_row(7, 2.0))                                   # B23
    # case 8 -- A9: B24 differs in one bucket bit of EVERY group (Hamming 6, and nearest) and so is
    # in none of A9's buckets: never a candidate (:72-78).  B25 alone is not enough.  No match.
    A.append(_row(8, 1.0))                                   # A9
    B.append(_row(8, 1.125, flips=(9, 19, 29, 39, 49, 59)))  # B24
    B.append(_row(8, 2.0))                                   # B25
    forward = [(2, 2), (3, 1), (4, 3), (5, 9), (6, 18), (8, 22)]
    # Backwards only B1 and B2 have two candidates (A2, A3), both -> A2; every other target sees a
    # single row of the first image.  Cross-check keeps (2 -> 2): reverse[2] = 2.  (3 -> 1) goes
    # because reverse[1] = 2.  The result is built in a vector that already holds one default
    # record per forward match (:187): six of them come first.
    expected = {False: (0, forward), True: (len(forward), [(2, 2)])}
    return np.stack(A), np.stack(B), expected
