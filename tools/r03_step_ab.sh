#!/bin/bash
# round 3: whole-step A/B of the level switches (timed-only bench lines)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03c
mkdir -p $out
cd $root
TTRAP_LEVEL_RECOMPUTE=0 python bench.py --timed-only --steps 10 --warmup 3 > $out/step_norecompute.json 2> $out/step_norecompute.err
TTRAP_LEVEL_RECOMPUTE=1 python bench.py --timed-only --steps 10 --warmup 3 > $out/step_recompute.json 2> $out/step_recompute.err
TTRAP_LEVEL_RECOMPUTE=0 TTRAP_WCONV_W3=1 python bench.py --timed-only --steps 10 --warmup 3 > $out/step_w3.json 2> $out/step_w3.err
for f in $out/step_*.json; do echo "== $f"; python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().split('\n')[-1])
print(d['ms_per_step'], d['value'], d.get('peak_memory_gb'))
"; done
tail -3 $out/*.err
