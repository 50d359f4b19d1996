#!/bin/bash
# round 5, session 15: C = 16 backward, strip kernel (default) against the per-stage pair with the halo-free dxw (TTRAP_WBWD1=0); trace with full torch kernel names
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_run15.txt
: > $out
for i in 1 2; do
  for v in "" "TTRAP_WBWD1=0"; do
    echo "== train step, $v" >> $out
    env $v python bench.py --timed-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
  done
done
cat $out
bash tools/r05_trace.sh r05_e > /dev/null
grep -n "at::native" gpurun_out/r05_e/bf16_kernel_trace.txt | cut -c1-300
