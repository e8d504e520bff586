"""Short randomised parity campaign (tools/fuzz_parity.py): random graphs, reads, score matrices and parameters over
every mode, GPU vs oracle.  The long runs are recorded in profiles/r01_notes.md."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_fuzz_parity_short(seed):
    # the second seed evaluates the path retirement every 16 records: the campaign's graphs are too small to retire anything
    # at the default period of 256
    env = dict(os.environ, RG_RETIRE_SHIFT="4") if seed == 12 else None
    if seed == 13:        # the i32 sweep forced, with its own path retirement evaluated every 16 records
        env = dict(os.environ, RG_RETIRE_SHIFT="4", RG_SWEEP_I32="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "20", str(seed)], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "failures 0" in r.stdout
