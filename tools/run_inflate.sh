timeout -k 10 300 python -m pytest tests/test_gpu_inflate.py -x -q > gpurun_out/r2_inf_tests.log 2>&1 || { echo "tests FAILED"; tail -5 gpurun_out/r2_inf_tests.log; exit 1; }
tail -1 gpurun_out/r2_inf_tests.log
timeout -k 10 300 python3 bench.py --copies 64 --steps 1 --warmup 0 --no-cpu-baseline --levels-64k 0 --verify 0 > gpurun_out/r2_inf_bench.log 2>&1 || { echo "bench FAILED"; tail -5 gpurun_out/r2_inf_bench.log; exit 1; }
python3 - <<'PY'
import json
for l in open("gpurun_out/r2_inf_bench.log"):
    if l.startswith("{"):
        d=json.loads(l); i=d["inflate"]; print("inflate", i["value"], i.get("ms_per_step"), i.get("roofline",{}).get("kernel_ms"), i.get("checked"))
PY
