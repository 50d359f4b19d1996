#!/bin/bash
# round 5, session 14: the two decodes of the same latents as one pass over 2 B clips (TimbreTrap.decode_pair)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_run14.txt
: > $out
python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -k "autocast or fp16 or reproducib or skip or oracle" > gpurun_out/r05_run14_model.log 2>&1; tail -3 gpurun_out/r05_run14_model.log >> $out
for i in 1 2; do
  for pg in 0 1; do
    echo "== train step, TTRAP_PAIR_DECODE=$pg" >> $out
    TTRAP_PAIR_DECODE=$pg python bench.py --timed-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f  peak GB %s' % (d['ms_per_step'], d.get('peak_memory_gb')))" >> $out
  done
done
cat $out
