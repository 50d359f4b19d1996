#!/bin/bash
# round 6: the strip kernel's ring with its own bank placement (rswz, conv_level_bf16.hip) against fswz (libttrap_rswz0.so = -DTT_BWDS_RSWZ=0):
# parity first, then isolated calls, PMC bank-conflict counters, and the train step.
cd "$(dirname "$0")/.."
python -m pytest tests/test_gpu_wide_bf16.py tests/test_gpu_multitile.py tests/test_gpu_determinism.py -q -m gpu -x -k "stagewise or level or multitile or bench_launch or reference_training_length or one_pass or riding" > gpurun_out/r06_rswz_tests.log 2>&1
tail -3 gpurun_out/r06_rswz_tests.log
for l in libttrap_rswz0.so libttrap_hip.so; do
  export TTRAP_LIB=$l
  echo "== $l"; KB_C=16 KB_D=1,2,3 KB_WHAT=bwd KB_N=10 python tools/kb_level.py 2>&1 | grep bwd
  KB_C=16 KB_D=1,2,3 KB_WHAT=bwd KB_N=3 bash tools/pmc_level.sh r06_$l > /dev/null 2>&1
  grep -A17 "k_wrb_bwds" gpurun_out/pmc_r06_$l/summary.txt | grep -E "k_wrb_bwds|BANK|IDX_ACTIVE|BUSY_CYCLES|WAIT_INST_LDS"
done
unset TTRAP_LIB
bash tools/ab.sh r06_rswz_ab -m train -r 3 -- "TTRAP_LIB=libttrap_rswz0.so" ""
