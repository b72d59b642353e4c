# inflate on config 4, both orders, per library variant; GPU inflate tests first
for v in "$@"; do
  ZSC_HIP_LIB=$PWD/build_variants/lib_$v.so timeout -k 10 200 python -m pytest tests/test_gpu_inflate.py -x -q > gpurun_out/i2_t_$v.log 2>&1 || { echo "$v tests FAILED"; tail -5 gpurun_out/i2_t_$v.log; exit 1; }
  ZSC_HIP_LIB=$PWD/build_variants/lib_$v.so timeout -k 10 300 python3 bench.py --copies 64 --steps 1 --warmup 0 --no-cpu-baseline --levels-64k 0 --verify 0 > gpurun_out/i2_$v.log 2>&1 || { echo "$v failed"; tail -3 gpurun_out/i2_$v.log; exit 1; }
  python3 - "$v" <<'PY'
import json,sys
v=sys.argv[1]
for l in open(f"gpurun_out/i2_{v}.log"):
    if l.startswith("{"):
        d=json.loads(l); i=d["inflate"]; print(v, "differ", i["value"], i["roofline"]["kernel_ms"], "identical", i["identical_neighbours"]["value"], i["identical_neighbours"]["kernel_ms"], i["all_ok"])
PY
done
