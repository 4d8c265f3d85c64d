"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (oracle/Makefile target `asan`; SURVEY.md section 5: the
reference has no sanitizer builds, the restatement it is checked against gets one).  A child interpreter with libasan
preloaded replays a whole small scenario -- Time-Surface raster, event selection, block matching, LM, fusion, clean,
regularisation, the wire decoder -- through the instrumented library; any heap / stack / bounds error or undefined operation
aborts it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import scenarios as S
from oracle import oracle as O
for name in ("upenn", "hkust"):
    sc = S.Scenario(name, n_ticks=3)
    m = O.OracleMapper(sc.params, sc.rig)
    for tk in sc.inputs():
        m.set_observation(tk["t"], tk["raw"][0], tk["raw"][1], tk["T"])
        m.set_poses(tk["stamps"], tk["poses"])
        m.tick(tk["ev"])
    assert len(m.get_map()) > 0
    m.tick_bm_only(tk["ev"])
a = open(%(root)r + "/tests/golden/wire_event_array.bin", "rb").read()
assert len(O.decode_event_array(a)[0]) == 7
for cut in range(0, len(a)):
    try:
        O.decode_event_array(a[:cut])
    except ValueError:
        pass
print("ASAN_CHILD_OK")
"""


def test_oracle_runs_clean_under_asan_and_ubsan():
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        import pytest
        pytest.skip("libasan.so not installed beside gcc")
    env = dict(os.environ, ESVO_ORACLE_ASAN="1", LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "ASAN_CHILD_OK" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-3000:]
