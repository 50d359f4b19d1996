#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03g
mkdir -p $out
cd $root
timeout 1500 python -m pytest tests/test_gpu_wide_bf16.py -x -q -k "stagewise or multitile or level_matches" > $out/pytest_wide.log 2>&1
tail -3 $out/pytest_wide.log
for v in 0 1; do
  for C in 32 16; do KB_ONLY=$C TTRAP_DXW=$v timeout 300 python tools/kb_wide.py 2>&1 | grep " bwd  " | sed "s/^/dxw=$v /"; done
  TTRAP_DXW=$v python bench.py --timed-only --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('dxw=$v step', d['ms_per_step'], d['value'])"
done
