# the round's closing measurements: the default bench line, the same under rocprofv3 --kernel-trace
# --stats (per-kernel totals exported by tools/kstats_db.py), and the GPU test suite
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/final_gpu_tests.log 2>&1 || { echo "gpu tests FAILED"; tail -5 gpurun_out/final_gpu_tests.log; exit 1; }
tail -1 gpurun_out/final_gpu_tests.log
timeout -k 10 500 python3 bench.py > gpurun_out/final_bench_default.json.log 2> gpurun_out/final_bench_default.err || { echo "bench FAILED"; tail -5 gpurun_out/final_bench_default.err; exit 1; }
tail -c 600 gpurun_out/final_bench_default.json.log; echo
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final_prof -o final -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/final_bench_under_rocprofv3.json.log 2> $R/gpurun_out/final_bench_under_rocprofv3.err || { echo "rocprof bench FAILED"; tail -5 $R/gpurun_out/final_bench_under_rocprofv3.err; exit 1; }
cd $R
python3 tools/kstats_db.py gpurun_out/final_prof/final_results.db gpurun_out/final_kernel_stats.csv | cut -c1-110 | head -14
ls gpurun_out/final_prof | head
rm -f gpurun_out/final_prof/*.db
