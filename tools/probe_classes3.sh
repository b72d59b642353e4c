# the three classes through the device plan (768 buffers of 512 KiB each), then the 1024-set headline:
#   gpurun -- 'bash tools/probe_classes3.sh'            (ZSC_HIP_LIB selects a build variant)
for k in text table bitmap; do
  timeout -k 10 120 python3 tools/probe_one.py $k 524288 768 2>&1 | grep -v amdgpu.ids || exit 1
done
timeout -k 10 300 python3 -m pytest tests/test_gpu_deflate.py -x -q -k "edge or canterbury or plan_large" 2>&1 | tail -2 || exit 1
bash tools/run_variants.sh default
