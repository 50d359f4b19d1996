#!/bin/bash
# round 5, fourth GPU session: A/B of the halo-free x tile in k_wrb_dxw (C = 32 per-stage backward) and of the C = 8 narrow split-operand shapes
cd "$(dirname "$0")/.."
root=$(pwd)
mkdir -p gpurun_out
out=gpurun_out/r05_run4.txt
: > $out
python -m pytest tests/test_gpu_wide_bf16.py tests/test_gpu_determinism.py tests/test_gpu_x3.py -q -m gpu > gpurun_out/r05_run4_tests.log 2>&1; tail -2 gpurun_out/r05_run4_tests.log >> $out
bash tools/build_variant.sh exp -DTTRAP_EXPERIMENTAL > /dev/null 2>&1
bash tools/build_variant.sh xa -DTTRAP_EXPERIMENTAL -DTT_X3N_R8=2 -DTT_X3N_MINW8=3 > /dev/null 2>&1
bash tools/build_variant.sh xc -DTTRAP_EXPERIMENTAL -DTT_X3N_R8=2 -DTT_X3N_MINW8=2 > /dev/null 2>&1
for v in 1 0 1 0; do
  echo "== train step, TTRAP_DXW_XH=$v (1: x tile with halo as in round 4; 0: halo-free)" >> $out
  TTRAP_LIB=libttrap_exp.so TTRAP_DXW_XH=$v python bench.py --timed-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
done
for v in 1 0; do
  echo "== isolated C = 32 backward calls, TTRAP_DXW_XH=$v" >> $out
  TTRAP_LIB=libttrap_exp.so TTRAP_DXW_XH=$v TTRAP_WBWD1=0 KB_C=32,16 KB_D=1,2,3 KB_WHAT=bwd KB_N=20 python tools/kb_level.py 2>&1 | grep bwd >> $out
done
for lib in exp xa xc exp xa xc; do
  echo "== inference configs[1], C = 8 narrow split-operand shape: $lib (exp: 4 rows / 2 per CU; xa: 2 rows / 3 per CU; xc: 2 rows / 2 per CU register cap)" >> $out
  TTRAP_LIB=libttrap_$lib.so python bench.py --mode infer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
done
echo "== xc with 3 workgroups per CU at C = 8 (registers 166-170)" >> $out
TTRAP_LIB=libttrap_xc.so TTRAP_X3N_PER_CU=3 python bench.py --mode infer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
cat $out
