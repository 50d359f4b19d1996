#!/bin/bash
# (round 5, session 23: the narrow fp32 strided layers with 2 columns per thread against 1 -- profiles/r05_valu4_ab.txt)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_run23.txt
: > $out
TTRAP_LIB=libttrap_valu2.so python -m pytest tests/test_gpu_conv.py tests/test_gpu_multitile.py -q -m gpu --tb=short > gpurun_out/r05_run23_tests.log 2>&1; tail -2 gpurun_out/r05_run23_tests.log >> $out
for i in 1 2; do
  for v in "TTRAP_LIB=libttrap_valu2.so" "TTRAP_LIB=libttrap_valu1.so"; do
    echo "== inference configs[1], $v" >> $out
    env $v python bench.py --mode infer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
  done
done
cat $out
