"""CPU-only: `python bench.py --gpus N` starts its own N rank processes when no launcher set WORLD_SIZE
(pairs are independent, brute_force_matcher.cc:46-61, so N ranks need nothing from each other but a
rendezvous), refuses N beyond the visible GPUs, and never prints a line for fewer ranks than asked."""
import importlib.util
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SFM_BENCH_BACKEND")}
    env.update(extra)
    return env


def test_more_ranks_than_gpus_is_refused_loudly():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "64"], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "--gpus 64" in r.stderr and "visible" in r.stderr
    assert r.stdout.strip() == ""            # no JSON line for a job that did not run


def test_world_size_that_disagrees_with_gpus_is_refused():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr and r.stdout.strip() == ""


def test_launcher_gives_every_rank_its_environment_and_relays_rank_zero(tmp_path, monkeypatch, capfd):
    probe = tmp_path / "probe.py"
    probe.write_text(
        "import os, sys\n"
        "r, w = os.environ['RANK'], os.environ['WORLD_SIZE']\n"
        "assert os.environ['LOCAL_RANK'] == r and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
        "print('rank %s of %s args %s' % (r, w, ' '.join(sys.argv[1:])), flush=True)\n")
    monkeypatch.setenv("SFM_BENCH_BACKEND", "gloo")      # ranks may outnumber GPUs (there are none here)
    bench = _bench_module()
    assert bench.launch_ranks(3, script=str(probe), argv=["--gpus", "3"]) == 0
    out, err = capfd.readouterr()
    assert out.strip() == "rank 0 of 3 args --gpus 3"             # stdout carries rank 0 only
    assert "rank 1 of 3" in err and "rank 2 of 3" in err


def test_launcher_fails_when_a_rank_fails_and_stops_the_others(tmp_path, monkeypatch):
    probe = tmp_path / "probe.py"
    probe.write_text(
        "import os, sys, time\n"
        "if os.environ['RANK'] == '1':\n"
        "    sys.exit(7)\n"
        "time.sleep(600)\n")
    monkeypatch.setenv("SFM_BENCH_BACKEND", "gloo")
    bench = _bench_module()
    import time
    t0 = time.monotonic()
    assert bench.launch_ranks(3, script=str(probe), argv=[]) == 7
    assert time.monotonic() - t0 < 60
