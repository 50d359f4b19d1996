#!/bin/bash
# round 5, third GPU session: parity of the re-cut narrow backward and of the narrow split-operand forward, their A/Bs, the inference trace
cd "$(dirname "$0")/.."
root=$(pwd)
mkdir -p gpurun_out
out=gpurun_out/r05_run3.txt
: > $out
python -m pytest tests/test_gpu_x3.py -q -m gpu > gpurun_out/r05_x3_tests.log 2>&1; tail -2 gpurun_out/r05_x3_tests.log >> $out
python -m pytest tests/test_gpu_wide_bf16.py tests/test_gpu_determinism.py -q -m gpu > gpurun_out/r05_wide_tests.log 2>&1; tail -2 gpurun_out/r05_wide_tests.log >> $out
python -m pytest tests/test_gpu_model.py -q -m gpu -k "beyond_2_31 or config1 or chunked or full_track or loss_scale or overflow" > gpurun_out/r05_model_tests.log 2>&1; tail -2 gpurun_out/r05_model_tests.log >> $out
for v in 0 1 0 1; do
  echo "== inference configs[1], TTRAP_X3N_INFER=$v" >> $out
  TTRAP_X3N_INFER=$v python bench.py --mode infer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f  value %.1f' % (d['ms_per_step'], d['value']))" >> $out
done
bash tools/build_variant.sh exp -DTTRAP_EXPERIMENTAL > /dev/null 2>&1
bash tools/build_variant.sh exp1 -DTTRAP_EXPERIMENTAL -DTT_NBF2_MINW4=1 > /dev/null 2>&1
for per in 3 4 5; do
  echo "== inference configs[1], x3n workgroups per CU = $per (C = 8 and C = 4 alike)" >> $out
  TTRAP_LIB=libttrap_exp.so TTRAP_X3N_PER_CU=$per python bench.py --mode infer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
done
for lib in exp exp1 exp exp1; do
  echo "== train step, narrow one-pass backward with launch bound $lib (exp: 4 workgroups / CU at C = 4, exp1: compiler's choice)" >> $out
  TTRAP_LIB=libttrap_$lib.so python bench.py --timed-only --no-cpu-baseline --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_infer -o t -- python3 $root/bench.py --mode infer --steps 6 --warmup 2 > /dev/null 2>&1
python3 $root/tools/prof_summary.py $(find /tmp/prof_infer -name 't_results.db') 8 > $root/gpurun_out/r05_infer_kernel_trace.txt
rm -rf /tmp/prof_infer
cd $root
head -30 gpurun_out/r05_infer_kernel_trace.txt | cut -c1-140 >> $out
cat $out
