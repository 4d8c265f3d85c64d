#!/bin/bash
# usage (GPU box): bash tools/fuzz_round.sh <first seed>  -- a batch of every randomised parity run on the current tree -> gpurun_out/fuzz_round_<seed>.txt
s=${1:-80000}
out=gpurun_out/fuzz_round_$s.txt
: > $out
for spec in "fuzz_parity 600 $s" "fuzz_dist 300 $((s+1000))" "fuzz_api 160 $((s+2000))" "fuzz_ts 300 $((s+3000))" "fuzz_track 200 $((s+4000))" "fuzz_sgm 100 $((s+5000))"; do
  set -- $spec
  echo "== tools/$1.py $2 $3" >> $out
  python tools/$1.py $2 $3 2>&1 | tail -4 >> $out
done
cat $out
