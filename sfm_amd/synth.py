"""Synthetic 128-d descriptors with planted correspondences.

Everything is derived from a counter-based splitmix64 stream and from integer
or correctly rounded IEEE operations only (add, multiply, divide, sqrt in
float64), so the same (seed, shape) gives the same bits on every machine; no
library random generator and no transcendental function is involved.

Rows imitate what the reference's extractor emits, RootSIFT
(src/feature_extractor/vlfeat_feature_extractor.cc:15-22: L1-normalise, then
element-wise sqrt): non-negative entries and unit L2 norm.  Purely random rows
are nearly equidistant, so the ratio test would reject almost everything;
correspondences are planted as noisy copies of shared rows.
"""
import numpy as np

DIM = 128
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, n, offset=0):
    """n 64-bit words of the splitmix64 stream `seed`, starting at word `offset`."""
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + i * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _stream(seed, tag):
    """Derive an independent stream seed from (seed, tag)."""
    return int(splitmix64(seed ^ (tag * 0xD1B54A32D192ED03 & 0xFFFFFFFFFFFFFFFF), 1)[0])


def rootsift_rows(n, seed):
    """n RootSIFT-like rows: x_k = sqrt(u_k / sum(u)), u_k integer in [1, 2^24]."""
    if n == 0:
        return np.zeros((0, DIM), np.float32)
    w = splitmix64(seed, n * DIM).reshape(n, DIM)
    u = (w >> np.uint64(40)).astype(np.int64) + 1
    s = u.sum(axis=1, keepdims=True)  # exact integer sum
    x = np.sqrt(u.astype(np.float64) / s.astype(np.float64))
    return x.astype(np.float32)


def noise(n, seed, sigma):
    """n x 128 zero-mean noise of standard deviation sigma (sum of four uniforms)."""
    w = splitmix64(seed, n * DIM).reshape(n, DIM)
    m = np.uint64(0xFFFF)
    s = ((w & m) + ((w >> np.uint64(16)) & m) + ((w >> np.uint64(32)) & m) + ((w >> np.uint64(48)) & m))
    # sum of four U[0,1) has variance 1/3: scale (s/65536 - 2) by sqrt(3)*sigma
    c = np.float64(1.7320508075688772) * np.float64(sigma)
    return (s.astype(np.float64) / 65536.0 - 2.0) * c


def permutation(n, seed):
    """A permutation of range(n): stable argsort of distinct-with-overwhelming-odds keys."""
    return np.argsort(splitmix64(seed, n), kind="stable")


def make_pair(n_a, n_b, planted, seed, sigma=0.01):
    """Two descriptor sets with `planted` true correspondences.

    Returns (A, B, truth) where truth[k] = (row in A, row in B).  B's planted
    rows are rows of A plus noise, the rest of both sets is independent.
    """
    planted = min(planted, n_a, n_b)
    A = rootsift_rows(n_a, _stream(seed, 1))
    B = rootsift_rows(n_b, _stream(seed, 2))
    src = permutation(n_a, _stream(seed, 3))[:planted]
    dst = permutation(n_b, _stream(seed, 4))[:planted]
    if planted:
        nz = noise(planted, _stream(seed, 5), sigma)
        B[dst] = (A[src].astype(np.float64) + nz).astype(np.float32)
    truth = np.stack([src, dst], axis=1).astype(np.int64) if planted else np.zeros((0, 2), np.int64)
    return A, B, truth


def make_image_set(n_images, n_features, shared_fraction, seed, sigma=0.01):
    """n_images descriptor sets that each contain noisy copies of a random
    `shared_fraction` of one common pool of n_features rows ("the scene"), the
    rest being independent rows.  Two images then share about
    shared_fraction^2 * n_features true correspondences.
    Returns a list of float32 arrays [n_features, 128].
    """
    pool = rootsift_rows(n_features, _stream(seed, 11))
    k = int(round(shared_fraction * n_features))
    out = []
    for img in range(n_images):
        s = _stream(seed, 100 + img)
        X = rootsift_rows(n_features, _stream(s, 1))
        if k:
            src = permutation(n_features, _stream(s, 2))[:k]
            dst = permutation(n_features, _stream(s, 3))[:k]
            X[dst] = (pool[src].astype(np.float64) + noise(k, _stream(s, 4), sigma)).astype(np.float32)
        out.append(X)
    return out


def scene_pool(planted, seed):
    """The rows every image of a scene set shares (see scene_image)."""
    return rootsift_rows(planted, _stream(seed, 12))


def scene_positions(img, n_features, planted, seed):
    """Row of image `img` that holds pool row r, for r < planted (needs no descriptor data)."""
    return permutation(n_features, _stream(_stream(seed, 200 + img), 3))[:planted]


def scene_image(img, n_features, pool, seed, sigma=0.01):
    """Image `img` of a scene set: a noisy copy of EVERY pool row at the rows scene_positions()
    names, independent rows elsewhere.  Any two images of the set therefore share exactly
    len(pool) true correspondences: row scene_positions(a)[r] of image a with row
    scene_positions(b)[r] of image b.  An image depends on (img, seed) only, so every process
    generates the images it needs and no others."""
    s = _stream(seed, 200 + img)
    X = rootsift_rows(n_features, _stream(s, 1))
    k = len(pool)
    if k:
        dst = scene_positions(img, n_features, k, seed)
        X[dst] = (pool.astype(np.float64) + noise(k, _stream(s, 4), sigma)).astype(np.float32)
    return X


def make_repetitive_set(n_images, n_features, shared_fraction, copies, seed, sigma=0.002):
    """Like make_image_set, but every shared scene row appears `copies` times in an image, each
    time with its own small noise (repeated structure, as the windows of a facade produce).  A
    query then has several near neighbours that agree in nearly all hash bits, which is what the
    cascade-hash matcher needs before it reports anything.
    Returns a list of float32 arrays [n_features, 128].
    """
    pool = rootsift_rows(n_features, _stream(seed, 11))
    k = int(round(shared_fraction * n_features / max(copies, 1)))
    out = []
    for img in range(n_images):
        s = _stream(seed, 100 + img)
        X = rootsift_rows(n_features, _stream(s, 1))
        if k:
            src = permutation(n_features, _stream(s, 2))[:k]
            dst = permutation(n_features, _stream(s, 3))[:k * copies]
            rows = np.repeat(pool[src].astype(np.float64), copies, axis=0)
            X[dst] = (rows + noise(k * copies, _stream(s, 4), sigma)).astype(np.float32)
        out.append(X)
    return out


def uniform(n, seed):
    """n doubles in [0, 1) from the splitmix64 stream."""
    return (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def make_two_view_matches(n, inlier_fraction, seed, noise=0.5, width=1600.0, height=1200.0):
    """n matches (lhs_xy, rhs_xy, is_inlier) between two views of one rigid scene: a fraction of
    them are projections of common 3-D points into two pinhole cameras (the second one shifted and
    turned about the vertical axis by fixed rational amounts: no trigonometric function involved),
    with uniform pixel noise of +-noise; the rest pair up unrelated positions.  Pixel coordinates,
    as the reference's key points are."""
    n_in = int(round(inlier_fraction * n))
    f = 1200.0
    cx, cy = width / 2.0, height / 2.0
    u = uniform(8 * n, _stream(seed, 31)).reshape(8, n)
    X = (u[0] - 0.5) * 8.0
    Y = (u[1] - 0.5) * 6.0
    Z = 6.0 + u[2] * 6.0
    lhs = np.stack([f * X / Z + cx, f * Y / Z + cy], axis=1)
    c, s = 0.96, 0.28                      # cos, sin of the turn: 0.96^2 + 0.28^2 = 1
    X2 = c * X + s * Z - 1.5
    Z2 = -s * X + c * Z + 0.4
    rhs = np.stack([f * X2 / Z2 + cx, f * (Y + 0.1) / Z2 + cy], axis=1)
    lhs += (u[3:5].T - 0.5) * 2.0 * noise
    rhs += (u[5:7].T - 0.5) * 2.0 * noise
    is_in = np.zeros(n, bool)
    order = permutation(n, _stream(seed, 32))
    is_in[order[:n_in]] = True
    w = uniform(2 * n, _stream(seed, 33)).reshape(n, 2)
    rhs[~is_in] = (w * np.array([width, height]))[~is_in]
    return lhs, rhs, is_in


def keypoints(n_features, image_id):
    """(x, y) = (feature index, image id): results can be checked from observations alone."""
    kp = np.empty((n_features, 2), np.float64)
    kp[:, 0] = np.arange(n_features, dtype=np.float64)
    kp[:, 1] = float(image_id)
    return kp
