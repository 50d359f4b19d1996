#!/bin/bash
# Final round-2 evidence on the GPU box (bf16 channels-last path = the bench default, fp32 path beside it).
# Writes under gpurun_out/r02g/ ; summaries produced locally by tools/prof_summary.py / tools/pmc_summary.py -> profiles/r02_g_*.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02g
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/bf16 -o t -- python3 $root/bench.py --timed-only --steps 6 --warmup 2 > $out/bench_bf16_timed.json 2> $out/bench_bf16_timed.err
rocprofv3 --kernel-trace --stats -d $out/fp32 -o t -- python3 $root/bench.py --precision fp32 --timed-only --steps 4 --warmup 2 > $out/bench_fp32_timed.json 2> $out/bench_fp32_timed.err
export KB_ONLY=32
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o p -- python3 $root/tools/kb_wide.py > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o p -- python3 $root/tools/kb_wide.py > $out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_ANY --kernel-trace -d $out/pmc_sq -o p -- python3 $root/tools/kb_wide.py > $out/pmc_sq.log 2>&1
cd $root
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_steps20.json 2> $out/bench_steps20.err
python3 bench.py --mode infer > $out/bench_infer.json 2> $out/bench_infer.err
ls $out
