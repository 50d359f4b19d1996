#!/bin/bash
# round 5, fifth GPU session: the bare top-of-loop barrier (no wait for the previous tile's stores) and the border-free staging path of the
# narrow forward, each against the library built without it; parity first
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_run5.txt
: > $out
python -m pytest tests/test_gpu_wide_bf16.py tests/test_gpu_determinism.py tests/test_gpu_x3.py tests/test_gpu_multitile.py -q -m gpu > gpurun_out/r05_run5_tests.log 2>&1; tail -2 gpurun_out/r05_run5_tests.log >> $out
python -m pytest tests/test_gpu_model.py -q -m gpu -k "autocast_bf16_step_matches or config1 or reproducibility or train_steps_match_oracle_mc2" > gpurun_out/r05_run5_model.log 2>&1; tail -2 gpurun_out/r05_run5_model.log >> $out
bash tools/build_variant.sh b0 -DTT_RAW_TOP_BARRIER=0 > /dev/null 2>&1
bash tools/build_variant.sh f0 -DTT_NCONV_FASTP=0 > /dev/null 2>&1
for lib in hip b0 f0 hip b0 f0; do
  echo "== train step, library $lib (hip: default; b0: __syncthreads() at the top of the tile loops; f0: narrow forward without the border-free staging path)" >> $out
  TTRAP_LIB=libttrap_$lib.so python bench.py --timed-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
done
for lib in hip b0 hip b0; do
  echo "== inference configs[1], library $lib" >> $out
  TTRAP_LIB=libttrap_$lib.so python bench.py --mode infer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
done
cat $out
