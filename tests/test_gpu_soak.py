"""A longer run of the lazy two-stream pipeline against the oracle: events staged tick by tick into a ring that wraps,
every tick's DepthMap fetched one tick late (esvo_map_get_committed) and compared with the canonical CPU oracle's
(tools/soak.py; the 120-tick / 60-tick runs quoted in DESIGN.md use the same script)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("args", [["upenn346x260", "3000", "40", "262144"], ["dsec640x480", "5000", "12", "2097152"],
                                  ["upenn346x260", "2000", "30", "262144", "resident"], ["dsec640x480", "4000", "8", "2097152", "resident"]])
def test_every_tick_of_a_long_run_equals_the_oracle(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py")] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " 0 ticks differ" in r.stdout
