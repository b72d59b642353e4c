# A/B of library variants built into build_variants/lib_NAME.so (e.g. make HIPFLAGS="... -DSG_W=9"; cp): the
# deflate headline per variant, through ZSC_HIP_LIB.  Edit the list, then: gpurun -- 'bash tools/run_variants.sh'
for v in ov256 ov1024; do
  ZSC_HIP_LIB=$PWD/build_variants/lib_$v.so timeout -k 10 150 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflate-streams 0 --levels-64k 0 --verify 0 > gpurun_out/var_$v.log 2>&1 || { echo "$v failed"; tail -3 gpurun_out/var_$v.log; break; }
  python3 - "$v" <<'PY'
import json,sys
v=sys.argv[1]
for l in open(f"gpurun_out/var_{v}.log"):
    if l.startswith("{"):
        d=json.loads(l); print(v, d["value"], d["ms_per_step"], d["roofline"].get("kernel_ms"))
PY
done
