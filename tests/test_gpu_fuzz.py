"""A short run of the randomised parity fuzz (tests/fuzz_gpu.py: random windows, switches, lag counts, '-' and N
bases; HIP path vs the C oracle, everything bit-exact).  Longer runs: python tests/fuzz_gpu.py <seconds> <seed>."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_windows_match_the_oracle():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_gpu.py"), "4", "11"], cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
