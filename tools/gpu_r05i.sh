#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
mkdir -p gpurun_out
export TMPDIR=/tmp
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05i_bench.json 2> gpurun_out/r05i_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05i_bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'sclk', d.get('sclk_mhz_timed_region'))
s=d.get('sustained') or {}
print('sustained', s.get('events_per_s'), s.get('ms_per_tick'), s.get('kernel_ms'), s.get('ms_per_tick_100tick_windows'))
p=d.get('parity') or {}
print('parity', p.get('oracle_equal'), (p.get('reference_node') or {}).get('iou'), (p.get('reference_node') or {}).get('rmse'))
print('roofline', json.dumps(d.get('roofline'))[:300])
print('cpu_baseline', json.dumps(d.get('cpu_baseline'))[:200])
print(list((d.get('other_operating_points') or {}).keys()))
PY
tail -3 gpurun_out/r05i_bench.err
