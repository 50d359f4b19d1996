#!/bin/bash
# round 3: fused wide backward + VALU trims -- parity, then per-launch timing at the bench shapes
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03b
mkdir -p $out
cd $root
timeout 1500 python -m pytest tests/test_gpu_wide_bf16.py -x -q > $out/pytest_wide.log 2>&1
tail -5 $out/pytest_wide.log
for C in 32 16 8 4; do
  KB_ONLY=$C timeout 300 python tools/kb_wide.py > $out/kb_C${C}.log 2>&1
done
KB_ONLY=32 TTRAP_WCONV_W3=1 timeout 300 python tools/kb_wide.py > $out/kb_C32_w3.log 2>&1
KB_ONLY=16 TTRAP_FBWD_TILE=1 timeout 300 python tools/kb_wide.py > $out/kb_C16_t1.log 2>&1
KB_ONLY=16 TTRAP_FBWD_PER_CU=2 timeout 300 python tools/kb_wide.py > $out/kb_C16_p2.log 2>&1
for f in $out/kb_*.log; do echo "== $f"; grep -h "bwd\|fwd" $f; done
