#!/bin/bash
# round 5, session 11: the level backward hands the transposed layer in front of it its gradient already gated (ops.GateLink,
# tt_wide_level_bwd_gated, tt_tconv16_bwd_pregated): parity, then the train step with / without (TTRAP_PREGATE)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_run11.txt
: > $out
python -m pytest tests/test_gpu_wide_bf16.py -q -m gpu -x > gpurun_out/r05_run11_tests.log 2>&1; tail -3 gpurun_out/r05_run11_tests.log >> $out
python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "autocast or fp16 or reproducib" > gpurun_out/r05_run11_model.log 2>&1; tail -3 gpurun_out/r05_run11_model.log >> $out
for i in 1 2; do
  for pg in 0 1; do
    echo "== train step, TTRAP_PREGATE=$pg" >> $out
    TTRAP_PREGATE=$pg python bench.py --timed-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
  done
done
cat $out
