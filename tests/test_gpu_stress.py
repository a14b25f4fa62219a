"""Short runs of the randomised confidence tools (tools/stress_*.py): random worlds, tiles and rays against the oracle.
A world may differ from the oracle only where the library REPORTS a loss (a documented capacity); anything else fails."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,seeds", [("stress_tick.py", 14), ("stress_broadphase.py", 16), ("stress_tiles.py", 10), ("stress_loopback.py", 8), ("stress_traffic.py", 6), ("stress_drawsort.py", 10), ("stress_lazy.py", 12)])
def test_randomised_worlds_against_the_oracle(tool, seeds):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "--seeds", str(seeds)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    print(r.stdout[-2500:], r.stderr[-1500:])
    assert r.returncode == 0, r.stdout[-800:]
