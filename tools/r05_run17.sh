#!/bin/bash
# round 5, session 17: narrow one-pass backward at C = 8 with three workgroups per CU: tiles of 12 (dilation 3: 8) rows instead of 16, the
# data-gradient weights read per tap from the LDS image instead of held in 72 registers (147-157 registers, cap 168)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_run17.txt
: > $out
for l in libttrap_n8.so libttrap_n8b.so; do
TTRAP_LIB=$l python -m pytest tests/test_gpu_wide_bf16.py -q -m gpu --tb=short -k "stagewise or level_backward or gated or multitile or capped or bench_" > gpurun_out/r05_run17_tests_$l.log 2>&1; tail -2 gpurun_out/r05_run17_tests_$l.log >> $out
done
for i in 1 2; do
  for v in "" "TTRAP_LIB=libttrap_n8.so" "TTRAP_LIB=libttrap_n8b.so"; do
    echo "== train step, $v" >> $out
    env $v python bench.py --timed-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
  done
done
cat $out
