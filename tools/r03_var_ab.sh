#!/bin/bash
# A/B of library variants (tools/build_variant.sh): strided kernels per call, then the whole step
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03f
mkdir -p $out
cd $root
for lib in "$@"; do
  tag=${lib%.so}
  TTRAP_LIB=$lib timeout 300 python tools/kb_stride.py > $out/kbs_$tag.log 2>&1
  TTRAP_LIB=$lib python bench.py --timed-only --steps 10 --warmup 3 > $out/step_$tag.json 2> $out/step_$tag.err
  echo "== $lib"; grep "C32\|C4 " $out/kbs_$tag.log
  python3 -c "
import json
d=json.loads(open('$out/step_$tag.json').read().strip().split('\n')[-1])
print('step', d['ms_per_step'], d['value'])"
done
