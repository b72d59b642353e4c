for v in "$@"; do
  ZSC_HIP_LIB=$PWD/build_variants/lib_$v.so timeout -k 10 300 python3 bench.py --copies 64 --steps 1 --warmup 0 --no-cpu-baseline --levels-64k 0 --verify 0 > gpurun_out/i2_$v.log 2>&1
  python3 - "$v" <<'PY'
import json,sys
v=sys.argv[1]
for l in open(f"gpurun_out/i2_{v}.log"):
    if l.startswith("{"):
        d=json.loads(l); i=d["inflate"]; print(v, "differ", i["roofline"]["kernel_ms"], "identical", i["identical_neighbours"]["kernel_ms"])
PY
done
