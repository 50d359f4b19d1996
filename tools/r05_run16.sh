#!/bin/bash
# round 5, session 16: the C = 16 strip backward with h1 straight from memory and the x rows staged one phase earlier (TT_BWDS_XE=1, the
# in-tree build) against the round-4 form (lib/libttrap_xe0.so, -DTT_BWDS_XE=0); parity first
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_run16.txt
: > $out
python -m pytest tests/test_gpu_wide_bf16.py tests/test_gpu_determinism.py -q -m gpu --tb=short -k "not fp16_build" > gpurun_out/r05_run16_tests.log 2>&1; tail -3 gpurun_out/r05_run16_tests.log >> $out
TTRAP_WBWD1=1 python -m pytest tests/test_gpu_wide_bf16.py -q -m gpu --tb=short -k "stagewise or level_backward or gated" > gpurun_out/r05_run16_tests1.log 2>&1; tail -3 gpurun_out/r05_run16_tests1.log >> $out
for i in 1 2; do
  for v in "" "TTRAP_LIB=libttrap_xe0.so"; do
    echo "== train step, $v" >> $out
    env $v python bench.py --timed-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
  done
done
cat $out
