"""The randomized soaks (tests/soak.py: every library entry point against the oracle on random inputs; tests/soak_cli.py:
`kmdiff-hip diff` on random run directories against the oracle's pipeline) for a fixed number of cases at a fixed seed (the same cases on every box) -- the tools
stay runnable, and every GPU test run adds a few dozen random cases to the fixed ones.  PARITY.md has what the long runs found."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,cases,word", [("soak.py", 24, "soak ok:"), ("soak_cli.py", 12, "cli soak ok:")])
def test_soak_tools_run_clean(tool, cases, word, tmp_path):
    env = dict(os.environ)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)          # (where a failing case would be saved)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", tool), "--cases", str(cases), "--seconds", "600", "--seed", "20261003"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert word in r.stdout
