#!/bin/bash
# round 5: the re-cut narrow one-pass backward (k_nrb_bwd_fused2) against the round-4 kernel and dispatch -- parity first, then the
# isolated calls, then inside the train step (same box, back to back).  TTRAP_NBF2 is a tt_tune switch: read by the -DTTRAP_EXPERIMENTAL build.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r05_nbf2_ab.txt
: > $out
python -m pytest tests/test_gpu_wide_bf16.py tests/test_gpu_determinism.py -q -m gpu -x -k "not fp16_build and not separate_kernel and not recompute_path" > gpurun_out/r05_nbf2_tests.log 2>&1
tail -3 gpurun_out/r05_nbf2_tests.log >> $out
python -m pytest tests/test_gpu_model.py tests/test_gpu_x3.py tests/test_gpu_wide_bf16.py -q -m gpu -k "beyond_2_31 or out_of_range or loss_scale or overflow or fp16_build or autocast_bf16_step_matches" > gpurun_out/r05_misc_tests.log 2>&1
tail -3 gpurun_out/r05_misc_tests.log >> $out
bash tools/build_variant.sh exp -DTTRAP_EXPERIMENTAL > /dev/null 2>&1 && export TTRAP_LIB=libttrap_exp.so
for v in 0 1; do
  echo "== isolated calls, TTRAP_NBF2=$v" >> $out
  TTRAP_NBF2=$v KB_C=4,8 KB_D=1,2,3 KB_WHAT=bwd KB_N=20 python tools/kb_level.py 2>&1 | grep bwd >> $out
done
for v in 0 1 0 1; do
  echo "== train step, TTRAP_NBF2=$v" >> $out
  TTRAP_NBF2=$v python bench.py --timed-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
done
cat $out
