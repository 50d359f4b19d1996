#!/bin/bash
# round 6 evidence on the GPU box: kernel traces of the bf16 train step, of the same step with skip connections (configs[4] model) and of
# inference configs[1] (summarised ON the box), the bench lines, and rocprofv3 --pmc passes of the kernels that are new or whose traffic figure
# was stale: the skip joins (tools/kb_skip.py) and the C = 16 strip backward (k_wrb_bwds<16>).
# Usage: tools/r06_profile.sh <tag>   -> gpurun_out/<tag>/, gpurun_out/pmc_r06_*/summary.txt
tag=${1:-r06p}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_bf16 -o t -- python3 $root/bench.py --timed-only --steps 6 --warmup 2 > $out/bench_bf16_timed.json 2> $out/bench_bf16_timed.err
python3 $root/tools/prof_summary.py $(find /tmp/prof_bf16 -name 't_results.db') 12 > $out/bf16_kernel_trace.txt
python3 $root/tools/prof_families.py $out/bf16_kernel_trace.txt > $out/bf16_families.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof_skip -o t -- python3 $root/bench.py --skip-connections --timed-only --steps 6 --warmup 2 > $out/bench_skip_timed.json 2> $out/bench_skip_timed.err
python3 $root/tools/prof_summary.py $(find /tmp/prof_skip -name 't_results.db') 12 > $out/skip_kernel_trace.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof_infer -o t -- python3 $root/bench.py --mode infer --steps 6 --warmup 2 > $out/bench_infer_timed.json 2> $out/bench_infer_timed.err
python3 $root/tools/prof_summary.py $(find /tmp/prof_infer -name 't_results.db') 8 > $out/infer_kernel_trace.txt
rm -rf /tmp/prof_bf16 /tmp/prof_skip /tmp/prof_infer
cd $root
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_steps20.json 2> $out/bench_steps20.err
python3 bench.py --mode infer > $out/bench_infer.json 2> $out/bench_infer.err
export KB_N=3 KB_D=1,2,3
KB_C=16 KB_WHAT=bwd bash tools/pmc_level.sh r06_C16 > /dev/null 2>&1
PMC_PY=tools/kb_skip.py bash tools/pmc_level.sh r06_skip > /dev/null 2>&1
ls gpurun_out/pmc_r06_*/summary.txt
head -40 $out/bf16_kernel_trace.txt | cut -c1-150
head -40 $out/skip_kernel_trace.txt | cut -c1-150
