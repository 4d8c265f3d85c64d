"""bench.py's `parity` block has teeth: the same command run on libesvo_hip_perturbed.so -- the product library built with
-DESVO_PERTURB_ONE_ULP, i.e. the depth of every eighth solver slot's point off by one unit in the last place (kernels_lm.hip) --
must report
`oracle_equal: false`, while the shipped library reports true.  (The twin is linked by __graft_entry__.build() and is never
loaded by the product: ESVO_HIP_LIB, the A/B switch of esvo_amd/lib.py, points this test's subprocess at it.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(env_extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "upenn346x260", "--steps", "3", "--warmup", "2",
           "--no-cpu-baseline", "--no-extras"]
    out = subprocess.run(cmd, env=dict(os.environ, **env_extra), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_one_flipped_bit_flips_oracle_equal():
    good = _line({})
    assert good["parity"]["oracle_equal"] is True, good["parity"]
    assert good["parity"]["oracle"]["map_size"] > 100
    # the in-run shader-clock probe delivered samples and a plausible clock (MI355X: 2.4 GHz peak)
    assert good["sclk_mhz_timed_region"] is not None and 500.0 < good["sclk_mhz_timed_region"] < 2600.0, good["sclk_mhz_timed_region"]
    twin = os.path.join(ROOT, "esvo_amd", "csrc", "libesvo_hip_perturbed.so")
    assert os.path.exists(twin), "run __graft_entry__.build() (it links the perturbed twin)"
    bad = _line({"ESVO_HIP_LIB": twin})
    assert bad["parity"]["oracle_equal"] is False, bad["parity"]
    assert bad["parity"]["oracle"]["map_size"] == good["parity"]["oracle"]["map_size"]   # one ulp: the same cells, other bits
