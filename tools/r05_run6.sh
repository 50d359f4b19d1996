#!/bin/bash
# round 5, sixth GPU session: the narrow split-operand blocks on the 16x16x16 matrix shape (parity, shape knobs), then the round's profile
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_run6.txt
: > $out
python -m pytest tests/test_gpu_x3.py -q -m gpu > gpurun_out/r05_run6_x3.log 2>&1; tail -2 gpurun_out/r05_run6_x3.log >> $out
python -m pytest tests/test_gpu_model.py -q -m gpu -k "config1 or chunked or full_track or beyond_2_31" > gpurun_out/r05_run6_model.log 2>&1; tail -2 gpurun_out/r05_run6_model.log >> $out
bash tools/build_variant.sh exp -DTTRAP_EXPERIMENTAL > /dev/null 2>&1
bash tools/build_variant.sh r1 -DTTRAP_EXPERIMENTAL -DTT_X3N_R8=1 -DTT_X3N_R4=1 > /dev/null 2>&1
bash tools/build_variant.sh r4 -DTTRAP_EXPERIMENTAL -DTT_X3N_R8=4 -DTT_X3N_R4=4 > /dev/null 2>&1
for cfg in "hip 0" "exp 3" "exp 4" "exp 5" "r1 4" "r4 3" "hip 0"; do
  set -- $cfg
  echo "== inference configs[1], library $1, x3n workgroups per CU $2 (0 = default: 3 at C = 8, 4 at C = 4; r1 / r4: 1 / 4 rows per step)" >> $out
  if [ "$2" = 0 ]; then TTRAP_LIB=libttrap_$1.so python bench.py --mode infer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
  else TTRAP_LIB=libttrap_$1.so TTRAP_X3N_PER_CU=$2 python bench.py --mode infer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out; fi
done
cat $out
bash tools/r05_profile.sh r05p
