"""Test helper: read/write the ".bin" SfMData layout (cereal binary archive of the reference's
serialize() functions, src/sfm_data_io.cc:13-47) independently of the C++ implementation."""
import struct

import numpy as np


def write_sfm_bin(path, images, keypoints=None, matches=None, width=0, height=0, view_paths=None):
    """images: {id: float32 [n,128]}; keypoints: {id: float64 [n,2]}; matches: {(a,b): (lhs_obs[n,2], rhs_obs[n,2], lhs_idx, rhs_idx)}."""
    keypoints = keypoints or {}
    matches = matches or {}
    with open(path, "wb") as f:
        f.write(struct.pack("<QQ", width, height))
        ids = sorted(images)
        f.write(struct.pack("<Q", len(ids)))
        for i in ids:
            p = (view_paths or {}).get(i, "image_%d.jpg" % i).encode()
            f.write(struct.pack("<qQ", i, len(p)) + p)
        f.write(struct.pack("<Q", len(keypoints)))
        for i in sorted(keypoints):
            k = np.ascontiguousarray(keypoints[i], dtype="<f8").reshape(-1, 2)
            f.write(struct.pack("<qQ", i, len(k)) + k.tobytes())
        f.write(struct.pack("<Q", len(ids)))
        for i in ids:
            d = np.ascontiguousarray(images[i], dtype="<f4")
            f.write(struct.pack("<qQ", i, d.size) + d.tobytes())
        f.write(struct.pack("<Q", len(matches)))
        for (a, b) in sorted(matches):
            lo, ro, li, ri = matches[(a, b)]
            n = len(li)
            f.write(struct.pack("<qqQ", a, b, n))
            rec = np.zeros(n, dtype=[("lx", "<f8"), ("ly", "<f8"), ("rx", "<f8"), ("ry", "<f8"), ("li", "<i8"), ("ri", "<i8")])
            if n:
                rec["lx"], rec["ly"] = np.asarray(lo)[:, 0], np.asarray(lo)[:, 1]
                rec["rx"], rec["ry"] = np.asarray(ro)[:, 0], np.asarray(ro)[:, 1]
                rec["li"], rec["ri"] = li, ri
            f.write(rec.tobytes())


def read_sfm_bin(path):
    buf = open(path, "rb").read()
    pos = 0

    def take(fmt):
        nonlocal pos
        v = struct.unpack_from(fmt, buf, pos)
        pos += struct.calcsize(fmt)
        return v if len(v) > 1 else v[0]

    out = {"views": {}, "key_points": {}, "descriptors": {}, "matches": {}}
    out["width"], out["height"] = take("<QQ")
    for _ in range(take("<Q")):
        i, n = take("<qQ")
        out["views"][i] = buf[pos:pos + n].decode()
        pos += n
    for _ in range(take("<Q")):
        i, n = take("<qQ")
        out["key_points"][i] = np.frombuffer(buf, "<f8", 2 * n, pos).reshape(n, 2).copy()
        pos += 16 * n
    for _ in range(take("<Q")):
        i, n = take("<qQ")
        out["descriptors"][i] = np.frombuffer(buf, "<f4", n, pos).reshape(-1, 128).copy()
        pos += 4 * n
    rec_t = np.dtype([("lx", "<f8"), ("ly", "<f8"), ("rx", "<f8"), ("ry", "<f8"), ("li", "<i8"), ("ri", "<i8")])
    for _ in range(take("<Q")):
        a, b, n = take("<qqQ")
        out["matches"][(a, b)] = np.frombuffer(buf, rec_t, n, pos).copy()
        pos += 48 * n
    assert pos == len(buf), "trailing bytes"
    return out
