for mb in 8192 4096 2048; do
  ZSC_HIP_SUBBATCH_MB=$mb timeout -k 10 150 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflate-streams 0 --levels-64k 0 --verify 0 > gpurun_out/sub_$mb.log 2>&1 || { echo "$mb failed"; tail -3 gpurun_out/sub_$mb.log; break; }
  python3 - "$mb" <<'PY'
import json,sys
v=sys.argv[1]
for l in open(f"gpurun_out/sub_{v}.log"):
    if l.startswith("{"):
        d=json.loads(l); print(v, d["value"], d["ms_per_step"], {k:d["config"].get(k) for k in d["config"] if "scratch" in k or "sub" in k}, d["roofline"].get("kernel_ms"))
PY
done
