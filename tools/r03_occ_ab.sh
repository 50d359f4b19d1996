#!/bin/bash
# round 3: A/B of the register-capped build (libttrap_occ.so) against the default, per kernel and whole step
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03d
mkdir -p $out
cd $root
export TTRAP_LEVEL_RECOMPUTE=0
for lib in libttrap_hip.so libttrap_occ.so; do
  tag=${lib%.so}
  TTRAP_LIB=$lib timeout 300 python tools/kb_wide.py > $out/kbw_$tag.log 2>&1
  TTRAP_LIB=$lib timeout 300 python tools/kb_stride.py > $out/kbs_$tag.log 2>&1
  TTRAP_LIB=$lib python bench.py --timed-only --steps 10 --warmup 3 > $out/step_$tag.json 2> $out/step_$tag.err
done
paste <(grep -h "fwd \|bwd " $out/kbw_libttrap_hip.log | grep -v fused) <(grep -h "fwd \|bwd " $out/kbw_libttrap_occ.log | grep -v fused | awk '{print $4}')
paste $out/kbs_libttrap_hip.log <(awk '{print $4}' $out/kbs_libttrap_occ.log)
for f in $out/step_*.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().split('\n')[-1])
print('$f', d['ms_per_step'], d['value'], d.get('peak_memory_gb'))
"; done
