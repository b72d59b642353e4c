# the round's closing measurements, all with the build in the tree (round 3):
#  1. the GPU test suite
#  2. the default bench line
#  3. the same under rocprofv3 --kernel-trace --stats (per-kernel totals exported by tools/kstats_db.py)
#  4. the deflate part alone under rocprofv3 (k_parse_seg's average next to the HIP-event time of the line)
#  5. SQ counters of k_parse_seg per class (tools/pmc_kernel.sh) and of the opt-in k_match_table on text
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/final_gpu_tests.log 2>&1 || { echo "gpu tests FAILED"; tail -5 gpurun_out/final_gpu_tests.log; exit 1; }
tail -1 gpurun_out/final_gpu_tests.log
timeout -k 10 600 python3 bench.py > gpurun_out/final_bench_default.json.log 2> gpurun_out/final_bench_default.err || { echo "bench FAILED"; tail -5 gpurun_out/final_bench_default.err; exit 1; }
echo "bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final_prof -o final -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/final_bench_under_rocprofv3.json.log 2> $R/gpurun_out/final_bench_under_rocprofv3.err || { echo "rocprof bench FAILED"; tail -5 $R/gpurun_out/final_bench_under_rocprofv3.err; exit 1; }
cd $R
python3 tools/kstats_db.py gpurun_out/final_prof/final_results.db gpurun_out/final_kernel_stats.csv > /dev/null
rm -rf gpurun_out/final_prof
echo "rocprof all done"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/defl_prof -o d -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflate-streams 0 --levels-64k 0 > $R/gpurun_out/final_deflate_under_rocprofv3.json.log 2>&1 || { echo "rocprof deflate failed"; exit 1; }
cd $R
python3 tools/kstats_db.py gpurun_out/defl_prof/d_results.db gpurun_out/final_deflate_kernel_stats.csv > /dev/null
rm -rf gpurun_out/defl_prof
echo "rocprof deflate done"
for k in text table bitmap; do
  bash tools/pmc_kernel.sh k_parse_seg $k > gpurun_out/final_pmc_sq_seg_$k.txt 2>&1
done
ZSC_HIP_TABLE=1 bash tools/pmc_kernel.sh k_match_table text > gpurun_out/final_pmc_sq_table_text.txt 2>&1
echo "pmc done"
