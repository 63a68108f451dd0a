"""A few seconds of tests/parity_stress.py per path, with fixed seeds: random sizes and kinds of inputs
against the CPU oracle, bit for bit (the tool stops with a non-zero exit at the first difference and
leaves the inputs in gpurun_out/parity_difference.npz).  The 4 x 3 case of
test_gpu_parity.py::test_four_rows_in_one_column_list_with_tied_keys was found this way."""
import argparse
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _tool():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "parity_stress.py")
    spec = importlib.util.spec_from_file_location("parity_stress", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("path,seed", [("brute", 101), ("batch", 102), ("cascade", 103), ("geometry", 104)])
def test_random_inputs_against_the_oracle(path, seed, capsys):
    tool = _tool()
    args = argparse.Namespace(path=path, seconds=4.0, seed=seed, max_log2=11.7)
    try:
        {"brute": tool.run_brute, "batch": tool.run_batch, "cascade": tool.run_cascade, "geometry": tool.run_geometry}[path](args)
    except SystemExit as e:     # the tool's way of reporting a difference
        pytest.fail("parity_stress --path %s --seed %d: %s\n%s" % (path, seed, e, capsys.readouterr().out[-2000:]))
    assert "identical to the oracle" in capsys.readouterr().out
