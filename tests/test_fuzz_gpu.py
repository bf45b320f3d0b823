"""A short slice of the randomised parity sweeps (scripts/fuzz_parity.py, scripts/fuzz_graph.py) inside the GPU
suite: random configurations of the HNSW path and of the graph path against the CPU oracle, fixed seeds, about a
minute in total.  The long sweeps are run by hand; DESIGN.md §5 records their results."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, seconds, seed):
    p = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "scripts", script), str(seconds), str(seed)],
                       capture_output=True, text=True, timeout=seconds + 300)
    tail = "\n".join(p.stdout.splitlines()[-15:])
    assert p.returncode == 0, tail + p.stderr[-2000:]
    last = p.stdout.strip().splitlines()[-1]
    assert last.startswith("done:") and " 0 bad" in last, tail
    return last


@pytest.mark.gpu
def test_hnsw_random_configurations_match_oracle(gpu):
    last = _run("fuzz_parity.py", 30, 11)
    assert int(last.split()[1]) >= 4, last  # it did get through a meaningful number of cases (heavy-delete cases are slow)
    assert " 0 delete refusals" in last, last


@pytest.mark.gpu
def test_graph_random_configurations_match_oracle(gpu):
    last = _run("fuzz_graph.py", 18, 11)
    assert int(last.split()[1]) >= 30, last
