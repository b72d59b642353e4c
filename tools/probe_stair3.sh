# per-class kernel times over staircase thresholds: gpurun -- 'bash tools/probe_stair3.sh 65 128 256'
for sm in "$@"; do
  for k in text table bitmap; do
    echo -n "stair_min $sm: "; ZSC_HIP_STAIR_MIN=$sm timeout -k 10 120 python3 tools/probe_one.py $k 524288 768 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
