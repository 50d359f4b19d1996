#!/bin/bash
# round 6: the wide forward kernels' LDS placement for ds_read_b128's real lane groups (libttrap_cswz1.so = -DTT_CSWZ_NEW=1) against the shipped one
cd "$(dirname "$0")/.."
TTRAP_LIB=libttrap_cswz1.so python -m pytest tests/test_gpu_wide_bf16.py tests/test_gpu_multitile.py -q -m gpu -x -k "stagewise or level or multitile or bench_launch or join" > gpurun_out/r06_cswz_tests.log 2>&1
tail -3 gpurun_out/r06_cswz_tests.log
for l in libttrap_hip.so libttrap_cswz1.so; do
  export TTRAP_LIB=$l
  echo "== $l"; KB_C=16,32 KB_D=1,2,3 KB_WHAT=fwd KB_N=10 python tools/kb_level.py 2>&1 | grep "fwd "
  KB_C=16,32 KB_D=1 KB_WHAT=fwd KB_N=3 bash tools/pmc_level.sh r06_$l > /dev/null 2>&1
  grep -A17 "k_wrb_conv" gpurun_out/pmc_r06_$l/summary.txt | grep -E "k_wrb_conv|BANK|IDX_ACTIVE|BUSY_CYCLES"
done
unset TTRAP_LIB
bash tools/ab.sh r06_cswz_ab -m train -r 3 -- "" "TTRAP_LIB=libttrap_cswz1.so"
