#!/bin/bash
# round 5 evidence on the GPU box: kernel traces of the bf16 train step and of inference configs[1] (summarised ON the box), the bench
# lines, and rocprofv3 --pmc passes (SQ x2, FETCH_SIZE, WRITE_SIZE -- each its own run) of the backward calls whose kernels changed this
# round: tt_wide_rb_bwd at C = 32 (k_wrb_bwd_a + k_wrb_dxw with the halo-free x tile) and at C = 8, 4 (k_nrb_bwd_fused, re-cut).
# Usage: tools/r05_profile.sh <tag>   -> gpurun_out/<tag>/, gpurun_out/pmc_r05_*/summary.txt
tag=${1:-r05p}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_bf16 -o t -- python3 $root/bench.py --timed-only --steps 6 --warmup 2 > $out/bench_bf16_timed.json 2> $out/bench_bf16_timed.err
python3 $root/tools/prof_summary.py $(find /tmp/prof_bf16 -name 't_results.db') 8 > $out/bf16_kernel_trace.txt
python3 $root/tools/prof_families.py $out/bf16_kernel_trace.txt > $out/bf16_families.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof_infer -o t -- python3 $root/bench.py --mode infer --steps 6 --warmup 2 > $out/bench_infer_timed.json 2> $out/bench_infer_timed.err
python3 $root/tools/prof_summary.py $(find /tmp/prof_infer -name 't_results.db') 8 > $out/infer_kernel_trace.txt
rm -rf /tmp/prof_bf16 /tmp/prof_infer
cd $root
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_steps20.json 2> $out/bench_steps20.err
python3 bench.py --mode infer > $out/bench_infer.json 2> $out/bench_infer.err
export KB_N=3 KB_D=1,2,3
KB_C=32 KB_WHAT=bwd bash tools/pmc_level.sh r05_C32 > /dev/null 2>&1
KB_C=8,4 KB_WHAT=bwd bash tools/pmc_level.sh r05_narrow > /dev/null 2>&1
ls gpurun_out/pmc_r05_*/summary.txt
head -50 $out/bf16_kernel_trace.txt | cut -c1-150
cat $out/bf16_families.txt
