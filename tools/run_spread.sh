# inflate on BASELINE config 4, replicas adjacent (spread 0) and neighbours all different (spread 2048), per library variant
for v in "$@"; do
for sp in 0 2048; do
  if [ $sp = 0 ]; then unset ZSC_HIP_INFLATE_SPREAD; else export ZSC_HIP_INFLATE_SPREAD=$sp; fi
  ZSC_HIP_LIB=$PWD/build_variants/lib_$v.so timeout -k 10 300 python3 bench.py --copies 64 --steps 1 --warmup 0 --no-cpu-baseline --levels-64k 0 --verify 0 > gpurun_out/spread_${v}_$sp.log 2>&1 || { echo "$v $sp failed"; tail -3 gpurun_out/spread_${v}_$sp.log; exit 1; }
  python3 - "$v" "$sp" <<'PY'
import json,sys
v,sp=sys.argv[1:3]
for l in open(f"gpurun_out/spread_{v}_{sp}.log"):
    if l.startswith("{"):
        d=json.loads(l); i=d["inflate"]; print(v, "spread", sp, "inflate", i["value"], i["roofline"]["kernel_ms"], i["all_ok"])
PY
done
done
