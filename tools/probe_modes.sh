# per-class kernel times over staircase thresholds, with and without the match table: bash tools/probe_modes.sh [copies]
N=${1:-768}
for k in text bitmap table; do
  for sm in 0 128 256 1024 100000; do
    ZSC_HIP_STAIR_MIN=$sm ZSC_HIP_NO_TABLE=1 ZSC_PROBE_CHILD=1 timeout -k 10 100 python3 tools/probe_table.py $k 524288 $N 2>&1 | grep -v amdgpu.ids | sed "s/cap=default/no table, stair_min $sm/"
  done
  ZSC_HIP_STAIR_MIN=256 ZSC_HIP_TABLE_CAP=8 ZSC_PROBE_CHILD=1 timeout -k 10 100 python3 tools/probe_table.py $k 524288 $N 2>&1 | grep -v amdgpu.ids | sed "s/cap=8/table cap 8, stair_min 256/"
done
