#!/bin/bash
# round 4, final evidence: the full GPU suite, the driver's bench command, the profile set r04_v3
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r04x_pytest.log 2>&1
tail -5 gpurun_out/r04x_pytest.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r04x_bench.json 2> gpurun_out/r04x_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04x_bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'sclk', d.get('sclk_mhz_timed_region'))
s=d.get('sustained') or {}
print('sustained', s.get('events_per_s'), s.get('ms_per_tick'), s.get('kernel_ms'))
print('parity', json.dumps(d.get('parity'))[:900])
print('roofline', json.dumps(d.get('roofline'))[:600])
o=d.get('other_operating_points',{})
for k,v in o.items():
    print(k, {kk: vv for kk, vv in v.items() if isinstance(vv,(int,float))})
PY
bash tools/profile_round.sh r04_v3 > /dev/null 2>&1
ls gpurun_out | grep r04_v3
