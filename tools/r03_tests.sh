#!/bin/bash
# round 3: full GPU test suite + smoke + default bench line
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03e
mkdir -p $out
cd $root
timeout 2400 python -m pytest tests -q -m gpu -x > $out/pytest_gpu.log 2>&1
tail -15 $out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $out/smoke.log
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('$out/bench_default.json').read().strip().split('\n')[-1])
print(d['ms_per_step'], d['value'], d['peak_memory_gb'])
print(json.dumps(d['roofline'])[:900])
print(json.dumps(d.get('roofline_fwd'))[:600])
print(d['inference_config1'])
for k,v in d['families'].items(): print(k, v)
"
