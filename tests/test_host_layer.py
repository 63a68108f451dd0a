"""Host C++ layer: file format, pair builder, interface plumbing (CPU) and the GPU driver."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from oracle import oracle
from sfm_amd import synth
from sfm_bin import read_sfm_bin, write_sfm_bin

HOST = os.path.join(ROOT, "sfm_amd", "host")


@pytest.fixture(scope="session")
def cpu_driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("host") / "cpu_driver")
    # SFM_TEST_CXXFLAGS: extra compiler flags for this program, e.g. "-fsanitize=address,undefined -g"
    cmd = ["g++", "-O2", "-std=c++17"] + os.environ.get("SFM_TEST_CXXFLAGS", "").split() + ["-I", HOST, "-o", exe, os.path.join(ROOT, "tests", "host", "cpu_driver.cc"),
           os.path.join(HOST, "exhaustive_pair_builder.cc"), os.path.join(HOST, "sfm_data_io.cc"),
           os.path.join(HOST, "sfm_data_json.cc"),
           "-x", "c", "-std=c11", "-ffp-contract=off", os.path.join(ROOT, "oracle", "bf_oracle.c"), "-lm", "-pthread"]
    subprocess.run(cmd, check=True)
    return exe


def two_image_set():
    A, B, _ = synth.make_pair(300, 260, 120, seed=31)
    images = {3: A, 8: B}
    kps = {i: synth.keypoints(len(d), i) for i, d in images.items()}
    return images, kps


def check_matches(data, images, cross):
    assert sorted(data["matches"]) == [(3, 8)]
    rec = data["matches"][(3, 8)]
    l, r, _ = oracle.match_pair(images[3], images[8], 0.8, cross)
    assert rec["li"].tolist() == l.tolist() and rec["ri"].tolist() == r.tolist()
    # keypoints were (feature index, image id): the observations must carry them over
    assert rec["lx"].tolist() == l.astype(float).tolist() and set(rec["ly"].tolist()) <= {3.0}
    assert rec["rx"].tolist() == r.astype(float).tolist() and set(rec["ry"].tolist()) <= {8.0}


def test_bin_roundtrip_is_byte_identical(cpu_driver, tmp_path):
    images, kps = two_image_set()
    l, r, _ = oracle.match_pair(images[3], images[8], 0.8, True)
    m = {(3, 8): (kps[3][l], kps[8][r], l, r)}
    src, dst = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    write_sfm_bin(src, images, kps, m, width=640, height=480)
    subprocess.run([cpu_driver, "--roundtrip", src, dst], check=True)
    assert open(src, "rb").read() == open(dst, "rb").read()


def test_load_rejects_unknown_extension_and_truncated_files(cpu_driver, tmp_path):
    images, kps = two_image_set()
    src = str(tmp_path / "a.bin")
    write_sfm_bin(src, images, kps)
    other = str(tmp_path / "a.dat")
    shutil.copy(src, other)
    assert subprocess.run([cpu_driver, "--roundtrip", other, str(tmp_path / "o.bin")]).returncode == 4
    cut = str(tmp_path / "cut.bin")
    open(cut, "wb").write(open(src, "rb").read()[:5000])
    assert subprocess.run([cpu_driver, "--roundtrip", cut, str(tmp_path / "o.bin")]).returncode == 4
    bad = str(tmp_path / "bad.json")
    open(bad, "w").write('{"value0": {"Image_Width": 1, "Views": [ {"key": 1, "value": ')
    assert subprocess.run([cpu_driver, "--roundtrip", bad, str(tmp_path / "o.bin")]).returncode == 4


def cereal_json_document(images, kps, matches, width, height):
    """The document cereal's JSONOutputArchive writes for the reference's serialize() functions
    (src/sfm_data_io.cc:13-47): unnamed values are value0, value1, ...; maps are arrays of
    key/value objects; pairs are first/second."""
    return {"value0": {
        "Image_Width": width, "Image_Height": height,
        "Views": [{"key": i, "value": {"value0": "img_%d.jpg" % i}} for i in sorted(images)],
        "KeyPoint": [{"key": i, "value": [{"x": float(x), "y": float(y)} for x, y in kps[i]]} for i in sorted(kps)],
        "Descriptors": [{"key": i, "value": {"value0": [float(v) for v in images[i].reshape(-1)]}} for i in sorted(images)],
        "Matches": [{"key": {"first": a, "second": b},
                     "value": [{"value0": float(lo[k][0]), "value1": float(lo[k][1]), "value2": float(ro[k][0]),
                                "value3": float(ro[k][1]), "value4": int(li[k]), "value5": int(ri[k])}
                               for k in range(len(li))]}
                    for (a, b), (lo, ro, li, ri) in sorted(matches.items())]}}


def test_json_archive_both_ways(cpu_driver, tmp_path):
    import json
    A, B, _ = synth.make_pair(40, 33, 20, seed=32)
    images = {3: A, 8: B, 11: A[:0]}
    kps = {i: synth.keypoints(len(d), i) + 0.25 for i, d in images.items()}
    l, r, _ = oracle.match_pair(A, B, 0.8, True)
    matches = {(3, 8): (kps[3][l], kps[8][r], l, r), (3, 11): (kps[3][:0], kps[11][:0], l[:0], r[:0])}
    doc = cereal_json_document(images, kps, matches, 640, 480)
    # a cereal-style document -> our reader -> binary archive: every value survives
    src = str(tmp_path / "from_cereal.json")
    json.dump(doc, open(src, "w"), indent=4)
    as_bin = str(tmp_path / "a.bin")
    subprocess.run([cpu_driver, "--roundtrip", src, as_bin], check=True)
    data = read_sfm_bin(as_bin)
    assert (data["width"], data["height"]) == (640, 480)
    assert all(np.array_equal(data["descriptors"][i].reshape(-1, 128), images[i]) for i in images)
    assert all(np.array_equal(np.asarray(data["key_points"][i]).reshape(-1, 2), kps[i]) for i in images)
    assert data["matches"][(3, 8)]["li"].tolist() == l.tolist() and data["matches"][(3, 8)]["ri"].tolist() == r.tolist()
    assert data["matches"][(3, 8)]["rx"].tolist() == kps[8][r][:, 0].tolist()
    assert len(data["matches"][(3, 11)]) == 0
    # binary archive -> our writer: a JSON document with cereal's structure and the same numbers
    back = str(tmp_path / "back.json")
    subprocess.run([cpu_driver, "--roundtrip", as_bin, back], check=True)
    got = json.load(open(back))
    assert got == doc
    # members in another order, and members we do not know, are tolerated
    odd = {"extra": [1, {"a": None}], "value0": dict(reversed(list(doc["value0"].items())), Comment="x")}
    odd_path = str(tmp_path / "odd.json")
    json.dump(odd, open(odd_path, "w"))
    again = str(tmp_path / "again.bin")
    subprocess.run([cpu_driver, "--roundtrip", odd_path, again], check=True)
    assert open(again, "rb").read() == open(as_bin, "rb").read()


@pytest.mark.parametrize("cross", [True, False])
def test_stage_driver_with_cpu_matcher(cpu_driver, tmp_path, cross):
    # BASELINE configs[0]: the matching stage on a 2-image set, CPU matcher, no GPU
    images, kps = two_image_set()
    path = str(tmp_path / "sfm_data.bin")
    write_sfm_bin(path, images, kps)
    out = subprocess.run([cpu_driver, path] + ([] if cross else ["--no-cross-check"]), check=True,
                         stdout=subprocess.PIPE, text=True).stdout
    assert "Load Sfm Data Finish" in out and "Need To Match 1 Pairs." in out
    data = read_sfm_bin(path)
    assert all(np.array_equal(data["descriptors"][i], images[i]) for i in images)
    check_matches(data, images, cross)


def test_missing_file_is_reported_not_fatal(cpu_driver, tmp_path):
    path = str(tmp_path / "nothing.bin")
    p = subprocess.run([cpu_driver, path], stdout=subprocess.PIPE, text=True)
    assert "Fails" in p.stdout and "Need To Match 0 Pairs." in p.stdout and p.returncode == 0
    assert read_sfm_bin(path)["matches"] == {}


@pytest.mark.gpu
@pytest.mark.parametrize("cross", [True, False])
def test_gpu_stage_driver(tmp_path, cross):
    exe = os.path.join(ROOT, "sfm_amd", "lib", "feature_match_main")
    assert os.path.exists(exe), "run python -m sfm_amd.build"
    images, kps = two_image_set()
    path = str(tmp_path / "sfm_data.bin")
    write_sfm_bin(path, images, kps)
    out = subprocess.run([exe, path] + ([] if cross else ["--no-cross-check"]), check=True,
                         stdout=subprocess.PIPE, text=True).stdout
    assert "Need To Match 1 Pairs." in out and "Finish 100.000000 %" in out
    check_matches(read_sfm_bin(path), images, cross)


@pytest.mark.gpu
@pytest.mark.parametrize("replication", ["host", "collective"])
def test_gpu_stage_driver_many_images(tmp_path, replication):
    """Both ways of getting the descriptors onto the GPU(s): uploads from the host, or every image to
    one GPU and an RCCL all-gather (SFM_HIP_REPLICATION, sfm_hip_replicate_images)."""
    exe = os.path.join(ROOT, "sfm_amd", "lib", "feature_match_main")
    imgs = synth.make_image_set(5, 400, 0.5, seed=77)
    images = {10 * i + 1: im[:n] for i, (im, n) in enumerate(zip(imgs, (400, 250, 129, 400, 1)))}
    kps = {i: synth.keypoints(len(d), i) for i, d in images.items()}
    path = str(tmp_path / "sfm_data.bin")
    write_sfm_bin(path, images, kps)
    subprocess.run([exe, path], check=True, stdout=subprocess.DEVNULL, env=dict(os.environ, SFM_HIP_REPLICATION=replication))
    data = read_sfm_bin(path)
    ids = sorted(images)
    assert sorted(data["matches"]) == [(a, b) for k, a in enumerate(ids) for b in ids[k + 1:]]
    for (a, b), rec in data["matches"].items():
        l, r, _ = oracle.match_pair(images[a], images[b], 0.8, True)
        assert rec["li"].tolist() == l.tolist() and rec["ri"].tolist() == r.tolist(), (a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["f32", "f16x3"])
def test_gpu_stage_driver_three_shards_on_one_gpu(tmp_path, engine):
    """The multi-GPU path of the C++ class (static shards, one host thread and context per shard,
    results written straight into the shared SfMData) with three shards forced onto the one GPU."""
    exe = os.path.join(ROOT, "sfm_amd", "lib", "feature_match_main")
    imgs = synth.make_image_set(7, 300, 0.5, seed=78)
    images = {i + 3: im[:n] for i, (im, n) in enumerate(zip(imgs, (300, 250, 0, 129, 300, 1, 257)))}
    kps = {i: synth.keypoints(len(d), i) for i, d in images.items()}
    path = str(tmp_path / "sfm_data.bin")
    write_sfm_bin(path, images, kps)
    env = dict(os.environ, SFM_HIP_OVERSUBSCRIBE="1")
    out = subprocess.run([exe, path, "--devices", "3", "--engine", engine], check=True, env=env,
                         stdout=subprocess.PIPE, text=True).stdout
    assert "Need To Match 21 Pairs." in out and out.count("\nFinish ") == 21 and "Finish 100.000000 %" in out
    data = read_sfm_bin(path)
    ids = sorted(images)
    assert sorted(data["matches"]) == [(a, b) for k, a in enumerate(ids) for b in ids[k + 1:]]
    for (a, b), rec in data["matches"].items():
        l, r, _ = oracle.match_pair(images[a], images[b], 0.8, True)
        assert rec["li"].tolist() == l.tolist() and rec["ri"].tolist() == r.tolist(), (a, b)
        # observations are (feature index, image id): src/sfm_data.hpp:17-25
        assert rec["lx"].tolist() == [float(v) for v in l] and set(rec["ry"].tolist()) <= {float(b)}


@pytest.mark.gpu
def test_gpu_stage_driver_eight_shards_ragged_images(tmp_path):
    """Eight shards (the partition an 8-GPU node gets: 2-D blocks of the pair triangle, cost-weighted)
    forced onto the one GPU, over images of very different sizes, two of them empty: every pair of the
    list comes back, each with the oracle's matches."""
    exe = os.path.join(ROOT, "sfm_amd", "lib", "feature_match_main")
    sizes = (600, 40, 0, 333, 600, 1, 257, 129, 512, 0, 64, 450, 90)
    imgs = synth.make_image_set(len(sizes), 600, 0.5, seed=81)
    images = {7 * i + 2: im[:n] for i, (im, n) in enumerate(zip(imgs, sizes))}
    kps = {i: synth.keypoints(len(d), i) for i, d in images.items()}
    path = str(tmp_path / "sfm_data.bin")
    write_sfm_bin(path, images, kps)
    env = dict(os.environ, SFM_HIP_OVERSUBSCRIBE="1")
    out = subprocess.run([exe, path, "--devices", "8"], check=True, env=env, stdout=subprocess.PIPE, text=True).stdout
    n_pairs = len(sizes) * (len(sizes) - 1) // 2
    assert "Need To Match %d Pairs." % n_pairs in out and out.count("\nFinish ") == n_pairs
    data = read_sfm_bin(path)
    ids = sorted(images)
    assert sorted(data["matches"]) == [(a, b) for k, a in enumerate(ids) for b in ids[k + 1:]]
    for (a, b), rec in data["matches"].items():
        l, r, _ = oracle.match_pair(images[a], images[b], 0.8, True)
        assert rec["li"].tolist() == l.tolist() and rec["ri"].tolist() == r.tolist(), (a, b)


def _cascade_expected(images, pairs, cross):
    from oracle import cascade
    whole, _, _ = cascade.match_set(images, pairs, cross)
    return whole


def _check_cascade_records(data, images, whole):
    assert sorted(data["matches"]) == sorted(whole)
    total = 0
    for (a, b), rec in data["matches"].items():
        nd, l, r = whole[(a, b)]
        # cascade_hash_matcher.cc:187: one value-initialised record per forward match comes first
        assert rec["li"].tolist() == [0] * nd + l.tolist(), (a, b)
        assert rec["ri"].tolist() == [0] * nd + r.tolist(), (a, b)
        # key points are (feature index, image id): observations of every record, the default ones
        # included, come from the key points its indices name (.cc:197-202)
        assert rec["lx"].tolist() == [float(v) for v in rec["li"]] and set(rec["ly"].tolist()) <= {float(a)}
        assert rec["rx"].tolist() == [float(v) for v in rec["ri"]] and set(rec["ry"].tolist()) <= {float(b)}
        total += len(l)
    return total


@pytest.mark.gpu
@pytest.mark.parametrize("cross", [True, False])
def test_gpu_stage_driver_cascade_matcher(tmp_path, cross):
    exe = os.path.join(ROOT, "sfm_amd", "lib", "feature_match_main")
    imgs = synth.make_repetitive_set(4, 900, 0.6, 2, seed=41, sigma=0.002)
    images = {5 * i + 2: im[:n] for i, (im, n) in enumerate(zip(imgs, (900, 700, 900, 333)))}
    kps = {i: synth.keypoints(len(d), i) for i, d in images.items()}
    path = str(tmp_path / "sfm_data.bin")
    write_sfm_bin(path, images, kps)
    out = subprocess.run([exe, path, "--matcher", "cascade"] + ([] if cross else ["--no-cross-check"]), check=True,
                         stdout=subprocess.PIPE, text=True).stdout
    ids = sorted(images)
    pairs = [(a, b) for k, a in enumerate(ids) for b in ids[k + 1:]]
    whole = _cascade_expected(images, pairs, cross)
    data = read_sfm_bin(path)
    assert _check_cascade_records(data, images, whole) > 100
    # the reference's lines: cascade_hash_matcher.cc:166, :171, :196
    assert "mean \n" in out and "Create Descriptors Finish\n" in out
    for (a, b) in pairs:
        assert "Piar : [%d,%d] %d Matches.\n" % (a, b, len(data["matches"][(a, b)])) in out


@pytest.mark.gpu
def test_gpu_cascade_three_shards_on_one_gpu(tmp_path):
    exe = os.path.join(ROOT, "sfm_amd", "lib", "feature_match_main")
    imgs = synth.make_repetitive_set(6, 500, 0.6, 2, seed=42, sigma=0.002)
    images = {i + 1: im[:n] for i, (im, n) in enumerate(zip(imgs, (500, 0, 500, 257, 1, 500)))}
    kps = {i: synth.keypoints(max(len(d), 1), i) for i, d in images.items()}
    path = str(tmp_path / "sfm_data.bin")
    write_sfm_bin(path, images, kps)
    env = dict(os.environ, SFM_HIP_OVERSUBSCRIBE="1")
    subprocess.run([exe, path, "--matcher", "cascade", "--devices", "3", "--quiet"], check=True, env=env)
    ids = sorted(images)
    pairs = [(a, b) for k, a in enumerate(ids) for b in ids[k + 1:]]
    whole = _cascade_expected(images, pairs, True)
    assert _check_cascade_records(read_sfm_bin(path), images, whole) > 50


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [1, 3])
def test_gpu_geometry_filter_program(tmp_path, devices):
    """GeometryFilter (src/sfm_main.cc:259-282) through the C++ drop-in: pairs with more than 30
    inliers keep exactly their inliers, the others disappear; one shard and three shards agree."""
    from oracle import geometry
    exe = os.path.join(ROOT, "sfm_amd", "lib", "geometry_filter_main")
    assert os.path.exists(exe), "run python -m sfm_amd.build"
    images = {i: synth.rootsift_rows(4, 50 + i) for i in (1, 2, 5, 9)}
    kps = {i: synth.keypoints(4, i) for i in images}
    spec = {(1, 2): (400, 0.8), (1, 5): (60, 0.1), (1, 9): (5, 1.0), (2, 5): (900, 0.6), (2, 9): (0, 1.0), (5, 9): (120, 0.9)}
    matches, views = {}, {}
    for k, (pair, (n, frac)) in enumerate(sorted(spec.items())):
        l, r, _ = synth.make_two_view_matches(n, frac, seed=300 + k)
        matches[pair] = (l, r, np.arange(n, dtype=np.int64), np.arange(n, dtype=np.int64)[::-1].copy())
        views[pair] = (l, r)
    path = str(tmp_path / "sfm_data.bin")
    write_sfm_bin(path, images, kps, matches)
    env = dict(os.environ, SFM_HIP_OVERSUBSCRIBE="1")
    out = subprocess.run([exe, path, "--seed", "77", "--devices", str(devices)], check=True, env=env,
                         stdout=subprocess.PIPE, text=True).stdout
    assert out.startswith("6Match \n") and out.count("inlier : ") == 6
    data = read_sfm_bin(path)
    kept = {}
    for k, pair in enumerate(sorted(spec)):
        l, r = views[pair]
        ok, _, flags, _ = geometry.compute_fundamental(l, r, seed=77, pair=k)
        assert "inlier : %d\n" % int(flags.sum()) in out
        if ok:
            kept[pair] = np.flatnonzero(flags)
    assert sorted(data["matches"]) == sorted(kept) and len(kept) >= 3
    for pair, idx in kept.items():
        rec = data["matches"][pair]
        l, r = views[pair]
        assert rec["li"].tolist() == idx.tolist()
        assert rec["ri"].tolist() == (len(l) - 1 - idx).tolist()
        assert rec["lx"].tolist() == l[idx, 0].tolist() and rec["ry"].tolist() == r[idx, 1].tolist()


@pytest.mark.gpu
def test_gpu_match_stream_hands_out_the_records_match_stores(tmp_path):
    """HipBruteForceMatcher::MatchStream (bounded memory: every finished pair's list goes to a sink and
    is not kept) against Match (brute_force_matcher.cc:80,83: everything into sfm_data.matches), on the
    same synthetic set, three shards on the one GPU: the same records pair by pair (a checksum over
    every record's indices and position), observations taken from the right key points, and nothing
    left in sfm_data.matches by the streaming form."""
    import json
    exe = os.path.join(ROOT, "sfm_amd", "lib", "match_set_bench")
    assert os.path.exists(exe), "run python -m sfm_amd.build"
    env = dict(os.environ, SFM_HIP_OVERSUBSCRIBE="1")
    lines = {}
    for sink in ("store", "discard"):
        out = subprocess.run([exe, "9", "700", "0.5", "3", "-1", "brute", sink], check=True, env=env,
                             stdout=subprocess.PIPE, text=True).stdout
        lines[sink] = json.loads(out.strip().splitlines()[-1])
    s, d = lines["store"], lines["discard"]
    assert s["pairs"] == d["pairs"] == 36 and s["matches"] == d["matches"] and s["matches"] >= 36 * 300
    assert s["records_checksum"] == d["records_checksum"]
    assert s["observation_mismatches"] == d["observation_mismatches"] == 0
    assert s["lists_in_sfm_data"] == 36 and d["lists_in_sfm_data"] == 0
    assert d["match_records"] == d["matches"] and d["gpus"] == 3
