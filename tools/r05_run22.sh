#!/bin/bash
# round 5, session 22: narrow fp32 strided / transposed layers (inference, fp32 train path) with four columns per thread (k_conv_valu4, in-tree)
# against one pixel per thread (lib/libttrap_valu1.so, -DTT_VALU4=0): parity, then inference configs[1]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_run22.txt
: > $out
python -m pytest tests/test_gpu_conv.py tests/test_gpu_multitile.py -q -m gpu --tb=short > gpurun_out/r05_run22_tests.log 2>&1; tail -3 gpurun_out/r05_run22_tests.log >> $out
python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -k "config1 or chunked or full_track or golden or train_steps_match_oracle and not mc2_full" > gpurun_out/r05_run22_model.log 2>&1; tail -3 gpurun_out/r05_run22_model.log >> $out
for i in 1 2; do
  for v in "" "TTRAP_LIB=libttrap_valu1.so"; do
    echo "== inference configs[1], $v" >> $out
    env $v python bench.py --mode infer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
  done
done
cat $out
