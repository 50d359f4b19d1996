#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03h
mkdir -p $out
cd $root
TTRAP_NARROW_FUSED16=0 timeout 1500 python -m pytest tests/test_gpu_wide_bf16.py -x -q -k "stagewise or multitile" > $out/pytest_wide_nofused.log 2>&1
tail -2 $out/pytest_wide_nofused.log
timeout 1500 python -m pytest tests/test_gpu_wide_bf16.py -x -q > $out/pytest_wide.log 2>&1
tail -2 $out/pytest_wide.log
for cfg in "1 1" "0 1" "0 0"; do set -- $cfg
  for C in 8 4; do KB_ONLY=$C TTRAP_NARROW_FUSED16=$1 TTRAP_NDXW=$2 timeout 300 python tools/kb_wide.py 2>&1 | grep " bwd  " | sed "s/^/fused16=$1 ndxw=$2 /"; done
done
