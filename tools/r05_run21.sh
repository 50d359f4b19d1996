#!/bin/bash
# round 5, session 21: strip backward at C = 16 / dilation 3 with the x image of its own too (three workgroups per CU instead of four)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_run21.txt
: > $out
for i in 1 2 3; do
  for v in "" "TTRAP_LIB=libttrap_xe3.so"; do
    echo "== train step, $v" >> $out
    env $v python bench.py --timed-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
  done
done
cat $out
