#!/bin/bash
# A/B of the one-pass wide backward (strip kernel k_wrb_bwds, workgroups per CU) against the per-stage kernels at the
# bench shape; env switches are read once per process
cd "$(dirname "$0")/.."
# the per-CU / tile knobs are tt_tune switches: compiled out of the shipped library, read by a -DTTRAP_EXPERIMENTAL build
bash tools/build_variant.sh exp -DTTRAP_EXPERIMENTAL > /dev/null 2>&1 && export TTRAP_LIB=libttrap_exp.so
out=gpurun_out/r04_bwd1_ab.txt
: > $out
for C in 32 16; do
  echo "== C=$C per-stage" >> $out
  TTRAP_WBWD1=0 KB_C=$C KB_D=1,2,3 KB_WHAT=bwd KB_N=20 python tools/kb_level.py 2>&1 | grep bwd >> $out
  for tile in 0; do for per in 2 3 4; do
    if [ $C = 32 ] && [ $per = 4 ]; then continue; fi
    echo "== C=$C strips TILE=$tile PER_CU=$per" >> $out
    TTRAP_BWDS_TILE=$tile TTRAP_BWDS_PER_CU=$per KB_C=$C KB_D=1,2,3 KB_WHAT=bwd1 KB_N=20 python tools/kb_level.py 2>&1 | grep bwd >> $out
  done; done
done
cat $out
