#!/bin/bash
# round 4 evidence on the GPU box: kernel traces of the bf16 (bench default) and fp32 steps, summarised ON the box (the databases
# stay in /tmp), plus the bench lines.  Usage: tools/r04_profile.sh <tag>   -> gpurun_out/<tag>/
tag=${1:-r04p}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_bf16 -o t -- python3 $root/bench.py --timed-only --steps 6 --warmup 2 > $out/bench_bf16_timed.json 2> $out/bench_bf16_timed.err
python3 $root/tools/prof_summary.py $(find /tmp/prof_bf16 -name 't_results.db') 8 > $out/bf16_kernel_trace.txt
python3 $root/tools/prof_families.py $out/bf16_kernel_trace.txt > $out/bf16_families.txt
if [ "$2" != "quick" ]; then
rocprofv3 --kernel-trace --stats -d /tmp/prof_fp32 -o t -- python3 $root/bench.py --precision fp32 --timed-only --steps 4 --warmup 2 > $out/bench_fp32_timed.json 2> $out/bench_fp32_timed.err
python3 $root/tools/prof_summary.py $(find /tmp/prof_fp32 -name 't_results.db') 6 > $out/fp32_kernel_trace.txt
python3 $root/tools/prof_families.py $out/fp32_kernel_trace.txt > $out/fp32_families.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof_infer -o t -- python3 $root/bench.py --mode infer --steps 6 --warmup 2 > $out/bench_infer_timed.json 2> $out/bench_infer_timed.err
python3 $root/tools/prof_summary.py $(find /tmp/prof_infer -name 't_results.db') 8 > $out/infer_kernel_trace.txt
cd $root
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_steps20.json 2> $out/bench_steps20.err
python3 bench.py --mode infer > $out/bench_infer.json 2> $out/bench_infer.err
fi
rm -rf /tmp/prof_bf16 /tmp/prof_fp32 /tmp/prof_infer
head -45 $out/bf16_kernel_trace.txt | cut -c1-150
cat $out/bf16_families.txt
