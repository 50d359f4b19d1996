#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
for v in 0 1; do
  echo "== TTRAP_COUT_V1=$v"
  TTRAP_COUT_V1=$v KB_WHAT=edge KB_C=4 timeout 300 python tools/kb_level.py 2>&1 | grep "convout fwd"
  TTRAP_COUT_V1=$v python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('step', round(d['ms_per_step'],2), [round(v['ms_per_step'],3) for k,v in d['families'].items() if isinstance(v,dict) and ('boundary' in k or 'CQT' in k)])"
done
timeout 300 python tools/kbench.py cqt 2>&1 | grep cqt
