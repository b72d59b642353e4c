for m in 18432 8192 3000; do
  ZSC_HIP_SEG_MIN=$m timeout -k 10 200 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflate-streams 0 --levels-64k 0 > gpurun_out/segmin_$m.log 2>&1 || { echo "$m failed"; tail -3 gpurun_out/segmin_$m.log; exit 1; }
  python3 - "$m" <<'PY'
import json,sys
v=sys.argv[1]
for l in open(f"gpurun_out/segmin_{v}.log"):
    if l.startswith("{"):
        d=json.loads(l); print(v, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"]["parse"], d["roofline"]["kernel_ms"]["parse_short"], d["checked"]["distinct_buffers_vs_oracle"])
PY
done
