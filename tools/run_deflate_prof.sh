R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflate-streams 0 --levels-64k 0 > gpurun_out/bal_bench.json.log 2>&1 || { echo failed; tail -3 gpurun_out/bal_bench.json.log; exit 1; }
python3 - <<'PY'
import json
for l in open("gpurun_out/bal_bench.json.log"):
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["roofline"]["launches_per_step"], d["roofline"]["kernel_ms"], d["config"].get("scratch_bytes"))
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/defl_prof -o d -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflate-streams 0 --levels-64k 0 > $R/gpurun_out/defl_under_rocprofv3.json.log 2>&1 || { echo "rocprof failed"; exit 1; }
cd $R
python3 tools/kstats_db.py gpurun_out/defl_prof/d_results.db gpurun_out/defl_kernel_stats.csv | cut -c1-12,100-170 | head -8
rm -f gpurun_out/defl_prof/*.db
