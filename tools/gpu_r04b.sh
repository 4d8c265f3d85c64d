#!/bin/bash
# round 4, second GPU session: the full GPU test suite, the default bench line again, deeper counters of the LM kernel
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
mkdir -p $out
cd $root
export ESVO_BENCH_STREAM_CACHE=/tmp/esvo_streams
( time timeout 1200 python -m pytest tests -m gpu -q > $out/r04b_pytest.log 2>&1 ) 2> $out/r04b_pytest.time
tail -8 $out/r04b_pytest.log
( time python bench.py > $out/r04b_bench.json 2> $out/r04b_bench.err ) 2> $out/r04b_bench.time
python - <<'P'
import json
d = json.load(open("gpurun_out/r04b_bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "sclk", d.get("sclk_mhz_timed_region"))
s = d.get("sustained") or {}
print("sustained", s.get("events_per_s"), s.get("ms_per_tick"), s.get("sclk_mhz"), s.get("sclk_mhz_per_xcd"))
print("parity", json.dumps(d.get("parity"))[:2500])
P
rocprofv3 -L > $out/r04b_counters_avail.txt 2>&1
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
G2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
G3="SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM"
G4="GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CU_CYCLES SQ_CYCLES"
bash tools/pmc_groups.sh r04b_overlap "ESVO_X=0" "$G1" "$G2" "$G3" "$G4" > /dev/null 2>&1
bash tools/pmc_groups.sh r04b_alone "ESVO_ONE_STREAM=1" "$G1" "$G2" "$G3" "$G4" > /dev/null 2>&1
head -70 $out/r04b_overlap_pmc.txt
