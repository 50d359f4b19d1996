#!/bin/bash
# A/B of library variants: CQT kernels, boundary kernels, then the whole step
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03g
mkdir -p $out
cd $root
for lib in "$@"; do
  tag=${lib%.so}
  echo "== $lib"
  TTRAP_LIB=$lib timeout 300 python tools/kbench.py cqt 2>&1 | grep -i "cqt\|fwd\|inv" | head -6
  TTRAP_LIB=$lib KB_WHAT=edge KB_C=4 timeout 300 python tools/kb_level.py 2>&1 | grep conv
  TTRAP_LIB=$lib python bench.py --timed-only --steps 10 --warmup 3 > $out/step_$tag.json 2> $out/step_$tag.err
  python3 -c "
import json
d=json.loads(open('$out/step_$tag.json').read().strip().split('\n')[-1])
print('step', d['ms_per_step'], d['value'])"
done
