#!/bin/bash
# round 4, last call: the full GPU suite, smoke(), and the driver's bench command on the final tree
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r05h_pytest.log 2>&1
tail -4 gpurun_out/r05h_pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05h_bench.json 2> gpurun_out/r05h_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05h_bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'sclk', d.get('sclk_mhz_timed_region'))
s=d.get('sustained') or {}
print('sustained', s.get('events_per_s'), s.get('ms_per_tick'), s.get('kernel_ms'), s.get('ms_per_tick_100tick_windows'))
p=d.get('parity') or {}
print('parity', p.get('oracle_equal'), (p.get('reference_node') or {}).get('iou'), (p.get('reference_node') or {}).get('rmse'))
print('roofline', json.dumps(d.get('roofline'))[:400])
PY
tail -3 gpurun_out/r05h_bench.err
