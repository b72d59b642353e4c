timeout -k 10 300 python -m pytest tests/test_gpu_deflate.py -x -q > gpurun_out/r2_k1_tests.log 2>&1 || { echo "tests FAILED"; tail -5 gpurun_out/r2_k1_tests.log; exit 1; }
tail -1 gpurun_out/r2_k1_tests.log
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/k1prof -o k1 -- python3 $GRAFT_REPO_ROOT/bench.py --copies 1024 --steps 2 --warmup 1 --no-cpu-baseline --inflate-streams 0 --levels-64k 0 --verify 0 > $GRAFT_REPO_ROOT/gpurun_out/k1prof.log 2>&1
echo rc=$?; cd $GRAFT_REPO_ROOT && python3 tools/kstats_db.py gpurun_out/k1prof/k1_results.db | cut -c1-100 | head -8
