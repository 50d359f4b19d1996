#!/bin/bash
# kernel trace of the bf16 train step only (summarised on the box).  Usage: tools/r05_trace.sh <tag>
tag=${1:-r05t}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_bf16 -o t -- python3 $root/bench.py --timed-only --steps 6 --warmup 2 > $out/bench_bf16_timed.json 2> $out/bench_bf16_timed.err
python3 $root/tools/prof_summary.py $(find /tmp/prof_bf16 -name 't_results.db') 8 > $out/bf16_kernel_trace.txt
python3 $root/tools/prof_families.py $out/bf16_kernel_trace.txt > $out/bf16_families.txt
rm -rf /tmp/prof_bf16
head -75 $out/bf16_kernel_trace.txt | cut -c1-150
cat $out/bf16_families.txt | grep -v " 0.0 ms"
