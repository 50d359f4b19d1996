#!/bin/bash
# A/B of the one-pass wide backward (k_wrb_bwd1) against the per-stage kernels at the bench shape; env switches are read once per process
cd "$(dirname "$0")/.."
out=gpurun_out/r04_bwd1_ab.txt
: > $out
for C in 32 16; do
  echo "== C=$C per-stage" >> $out
  TTRAP_WBWD1=0 KB_C=$C KB_D=1,2,3 KB_WHAT=bwd KB_N=20 python tools/kb_level.py >> $out 2>&1
  for tile in 0 1; do for per in 2 3; do
    echo "== C=$C one-pass TILE=$tile PER_CU=$per" >> $out
    TTRAP_BWD1_TILE=$tile TTRAP_BWD1_PER_CU=$per KB_C=$C KB_D=1,2,3 KB_WHAT=bwd1 KB_N=20 python tools/kb_level.py >> $out 2>&1
  done; done
done
cat $out
