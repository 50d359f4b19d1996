#!/bin/bash
# round 5, eighth GPU session: narrow split-operand blocks, hybrid matrix shapes (C = 4: 4x4x4; C = 8: K = 16 main + K = 32 cross term);
# the narrow one-pass backward with exact-size LDS images (C = 4 / dilation 3: a fourth workgroup per CU)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_run8.txt
: > $out
python -m pytest tests/test_gpu_x3.py tests/test_gpu_wide_bf16.py tests/test_gpu_determinism.py -q -m gpu > gpurun_out/r05_run8_tests.log 2>&1; tail -2 gpurun_out/r05_run8_tests.log >> $out
python -m pytest tests/test_gpu_model.py -q -m gpu -k "config1 or chunked or full_track or autocast_bf16_step_matches" > gpurun_out/r05_run8_model.log 2>&1; tail -2 gpurun_out/r05_run8_model.log >> $out
python tools/kb_x3n.py >> $out 2>&1
for i in 1 2; do
  echo "== inference configs[1]" >> $out
  python bench.py --mode infer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
  echo "== train step" >> $out
  python bench.py --timed-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
done
KB_C=4 KB_D=1,2,3 KB_WHAT=bwd KB_N=20 python tools/kb_level.py 2>&1 | grep bwd >> $out
cat $out
