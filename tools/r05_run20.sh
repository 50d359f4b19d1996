#!/bin/bash
# round 5, session 20: squared-error losses with the gradient written in the forward pass (ops.LOSS_FUSED), TTRAP_LOSS_FUSED = 0 / 1
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_run20.txt
: > $out
python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -k "losses or consistency or golden or autocast_bf16_step_matches or train_steps_match_oracle and not mc2" > gpurun_out/r05_run20_model.log 2>&1; tail -3 gpurun_out/r05_run20_model.log >> $out
for i in 1 2; do
  for v in 0 1; do
    echo "== train step, TTRAP_LOSS_FUSED=$v" >> $out
    TTRAP_LOSS_FUSED=$v python bench.py --timed-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f  peak GB %.2f' % (d['ms_per_step'], d.get('peak_memory_gb') or 0))" >> $out
  done
done
cat $out
