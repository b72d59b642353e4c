# per-class kernel times of build variants: gpurun -- 'bash tools/probe_variants3.sh NAME...'
for v in "$@"; do
  if [ "$v" = default ]; then unset ZSC_HIP_LIB; else export ZSC_HIP_LIB=$PWD/build_variants/lib_$v.so; fi
  for k in text table bitmap; do
    echo -n "$v: "; timeout -k 10 120 python3 tools/probe_one.py $k 524288 768 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
