#!/bin/bash
# round 4, third GPU session: the tests that failed in session B + the new tracker tests, the closed loop, spatial-partition probes
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
timeout 900 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_multiproc.py tests/test_track_normal.py tests/test_gpu_closed_loop.py tests/test_gpu_track.py tests/test_cpp_host.py -m gpu -q > $out/r04c_pytest.log 2>&1
tail -6 $out/r04c_pytest.log
python - <<'P' > $out/r04c_closed_loop.txt 2>&1
import numpy as np
from esvo_amd import closed_loop as cl
r = cl.run(n_ticks=15)
med = lambda v: float(np.median(np.asarray(v[3:])))
print("cycle", med(r["cycle_ms"]), "track", med(r["track_ms"]), "map", med(r["map_ms"]), "final err mm", r["pos_err"][-1] * 1e3, "points", int(np.median(r["points"])))
P
cat $out/r04c_closed_loop.txt
timeout 900 python tools/bound_probe.py dsec640x480 30 3 base,two_lm,cu_back32,cu_back16_two_lm,cu_back32_two_lm,cu_back48_two_lm,cu_back64_two_lm,cu_back32_front32_two_lm,cu_back32_front16_two_lm,base_third > $out/r04c_bound.json 2> $out/r04c_bound.txt
cat $out/r04c_bound.txt
