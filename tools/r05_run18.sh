#!/bin/bash
# round 5, session 18: split / join of the split-operand kernels with the mixed-precision fma spelled out (TT_X3_MIX=1, in-tree) against the
# compiler's packed-fp32 form (lib/libttrap_mix0.so): parity, then inference configs[1]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_run18.txt
: > $out
python -m pytest tests/test_gpu_x3.py -q -m gpu --tb=short > gpurun_out/r05_run18_tests.log 2>&1; tail -3 gpurun_out/r05_run18_tests.log >> $out
python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -k "config1 or chunked or full_track or transcribe or inference or one_shot" > gpurun_out/r05_run18_model.log 2>&1; tail -3 gpurun_out/r05_run18_model.log >> $out
for i in 1 2; do
  for v in "" "TTRAP_LIB=libttrap_mix0.so"; do
    echo "== inference configs[1], $v" >> $out
    env $v python bench.py --mode infer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
  done
done
cat $out
