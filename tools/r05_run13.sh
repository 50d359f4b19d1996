#!/bin/bash
# round 5, session 13: gate links around the latent heads (tt_latent16_expand_gated, tt_latent16_*_pregated, tt_tconv16_bwd_pregated gate_dx)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_run13.txt
: > $out
python -m pytest tests/test_gpu_wide_bf16.py tests/test_gpu_determinism.py -q -m gpu --tb=short > gpurun_out/r05_run13_tests.log 2>&1; tail -3 gpurun_out/r05_run13_tests.log >> $out
python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -k "autocast or fp16 or reproducib or skip" > gpurun_out/r05_run13_model.log 2>&1; tail -3 gpurun_out/r05_run13_model.log >> $out
for i in 1 2; do
  for pg in 0 1; do
    echo "== train step, TTRAP_PREGATE=$pg" >> $out
    TTRAP_PREGATE=$pg python bench.py --timed-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
  done
done
cat $out
