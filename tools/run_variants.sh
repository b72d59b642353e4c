# A/B of library variants built into build_variants/lib_NAME.so (e.g. hipcc ... -DSG_W=10): the deflate
# headline per variant, through ZSC_HIP_LIB.  gpurun -- 'bash tools/run_variants.sh NAME...'
for v in "$@"; do
  if [ "$v" = default ]; then unset ZSC_HIP_LIB; else export ZSC_HIP_LIB=$PWD/build_variants/lib_$v.so; fi
  timeout -k 10 200 python3 bench.py --copies 1024 --seeds 8 --steps 2 --warmup 1 --no-cpu-baseline --inflate-streams 0 --levels-64k 0 --verify 0 > gpurun_out/var_$v.log 2>&1 || { echo "$v failed"; tail -3 gpurun_out/var_$v.log; break; }
  python3 - "$v" <<'PY'
import json,sys
v=sys.argv[1]
for l in open(f"gpurun_out/var_{v}.log"):
    if l.startswith("{"):
        d=json.loads(l); print(v, d["value"], d["ms_per_step"], {k: d["roofline"]["kernel_ms"][k] for k in ("hash_sort","match_table","parse","huff_plan")})
PY
done
