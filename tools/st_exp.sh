root=$GRAFT_REPO_ROOT
cd $root
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_wire_golden.py -m gpu -q -x -k "time_surface or wire or fullsize" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for w in dsec640x480 upenn346x260 hd1280x720; do
out=$root/gpurun_out/sa_cur; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o t -- python $root/tools/standalone_kernels.py $w 10 > /dev/null 2>&1
echo "== cur $w"; python $root/tools/prof_summary.py $out/t_results.db $out/ks.csv > /dev/null; grep -E "ts_|gaussian" $out/ks.csv | sed 's/(.*)"//'
rm -rf $out
done
cd $root
bash tools/ab_bench.sh 2 cur base
