# SQ counters of one kernel of the deflate plan on one class: `bash tools/pmc_kernel.sh k_match_table text`
# (768 copies of a 512 KiB buffer = 403 MB per dispatch, 3 dispatches; two counter passes)
KERN=${1:-k_match_table}
KIND=${2:-text}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="$R/tools/probe_one.py $KIND 524288 768"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $R/gpurun_out/pmc_k_a -o a -- python3 $ARGS > $R/gpurun_out/pmc_k_a.log 2>&1 || { echo "pass a failed"; tail -3 $R/gpurun_out/pmc_k_a.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM --kernel-trace -d $R/gpurun_out/pmc_k_b -o b -- python3 $ARGS > $R/gpurun_out/pmc_k_b.log 2>&1 || { echo "pass b failed"; tail -3 $R/gpurun_out/pmc_k_b.log; exit 1; }
cd $R
tail -1 gpurun_out/pmc_k_a.log
python3 tools/pmc_db.py gpurun_out/pmc_k_a/a_results.db $KERN 402653184
python3 tools/pmc_db.py gpurun_out/pmc_k_b/b_results.db $KERN 402653184
rm -rf gpurun_out/pmc_k_a gpurun_out/pmc_k_b
