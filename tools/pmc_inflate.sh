# SQ counters of k_inflate on a slice of BASELINE config 4 (two passes), summed by tools/pmc_db.py
cd /tmp && export TMPDIR=/tmp
ORDER=${1:-identical}   # differ | identical
R=$GRAFT_REPO_ROOT
ARGS="$R/bench.py --copies 8 --steps 1 --warmup 0 --inflate-streams 131072 --inflate-order $ORDER --levels-64k 0 --no-cpu-baseline --verify 0"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $R/gpurun_out/pmc_inf_a -o a -- python3 $ARGS > $R/gpurun_out/pmc_inf_a.log 2>&1 || { echo "pass a failed"; tail -3 $R/gpurun_out/pmc_inf_a.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM --kernel-trace -d $R/gpurun_out/pmc_inf_b -o b -- python3 $ARGS > $R/gpurun_out/pmc_inf_b.log 2>&1 || { echo "pass b failed"; tail -3 $R/gpurun_out/pmc_inf_b.log; exit 1; }
cd $R
OUT=$(python3 -c "
import json
for l in open('gpurun_out/pmc_inf_a.log'):
    if l.startswith('{'): print(json.loads(l)['inflate']['output_bytes'])
" 2>/dev/null)
echo "out_bytes per dispatch: $OUT"
python3 tools/pmc_db.py gpurun_out/pmc_inf_a/a_results.db k_inflate ${OUT:-0}
python3 tools/pmc_db.py gpurun_out/pmc_inf_b/b_results.db k_inflate ${OUT:-0}
rm -rf gpurun_out/pmc_inf_a gpurun_out/pmc_inf_b
