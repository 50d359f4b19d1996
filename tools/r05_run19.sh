#!/bin/bash
# round 5, session 19: the gate link between Encoder.convin and the first level (tt_convin16_bwd with y == NULL)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_run19.txt
: > $out
python -m pytest tests/test_gpu_wide_bf16.py -q -m gpu --tb=short -k "edge_convs or embeddings_stay or channels_last_end or fp16_build" > gpurun_out/r05_run19_tests.log 2>&1; tail -3 gpurun_out/r05_run19_tests.log >> $out
python -m pytest tests/test_gpu_model.py -q -m gpu --tb=short -k "autocast or fp16 or skip" > gpurun_out/r05_run19_model.log 2>&1; tail -3 gpurun_out/r05_run19_model.log >> $out
for i in 1 2 3; do
    echo "== train step" >> $out
    python bench.py --timed-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
done
cat $out
