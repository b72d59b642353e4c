# HBM traffic counters (FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace only) of the parse and
# inflate kernels on a slice of the bench workloads; sums per kernel by tools/pmc_db.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="$R/bench.py --copies 512 --steps 1 --warmup 0 --inflate-streams 131072 --inflate-order differ --levels-64k 0 --no-cpu-baseline --verify 0"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_hbm_$c -o h -- python3 $ARGS > $R/gpurun_out/pmc_hbm_$c.log 2>&1 || { echo "$c failed"; tail -3 $R/gpurun_out/pmc_hbm_$c.log; exit 1; }
done
cd $R
python3 - <<'PY'
import json
for l in open("gpurun_out/pmc_hbm_FETCH_SIZE.log"):
    if l.startswith("{"):
        d = json.loads(l)
        print("deflate input bytes per step", d["config"]["input_bytes_per_gpu"], "compressed", d["config"]["compressed_bytes_total"], "launches", d["roofline"]["launches_per_step"])
        print("inflate output bytes per dispatch", d["inflate"]["output_bytes"], "compressed", d["inflate"]["compressed_bytes"])
PY
for c in FETCH_SIZE WRITE_SIZE; do
  python3 tools/pmc_db.py gpurun_out/pmc_hbm_$c/h_results.db k_parse_seg
  python3 tools/pmc_db.py gpurun_out/pmc_hbm_$c/h_results.db k_inflate
  python3 tools/pmc_db.py gpurun_out/pmc_hbm_$c/h_results.db k_hash_sort
  python3 tools/pmc_db.py gpurun_out/pmc_hbm_$c/h_results.db k_link_prev
done
rm -rf gpurun_out/pmc_hbm_FETCH_SIZE gpurun_out/pmc_hbm_WRITE_SIZE
