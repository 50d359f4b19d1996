#!/bin/bash
# per-family step breakdown of library variants
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03g
mkdir -p $out
cd $root
for lib in "$@"; do
  tag=${lib%.so}
  TTRAP_LIB=$lib python bench.py --no-cpu-baseline > $out/fam_$tag.json 2> $out/fam_$tag.err
  python3 -c "
import json
d=json.loads(open('$out/fam_$tag.json').read().strip().split('\n')[-1])
print('== $lib step', round(d['ms_per_step'],2))
for k,v in d['families'].items():
    if isinstance(v,dict): print('  %-90s %.3f' % (k[:90], v['ms_per_step']))"
done
