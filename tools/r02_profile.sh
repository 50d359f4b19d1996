#!/bin/bash
# Round-2 evidence run on the GPU box: kernel trace of the bench, PMC passes of the roofline kernel, CQT traces.
# Writes under gpurun_out/r02p/ ; summaries are produced locally by tools/prof_summary.py / tools/pmc_summary.py.
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r02p
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/bench -o b -- python3 $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $out/bench.log 2>&1
export KB_ITERS=3
KB_C=32 rocprofv3 --kernel-trace --stats -d $out/cqt -o c -- python3 $root/tools/kbench.py cqt > $out/cqt.log 2>&1
export KB_C=32 KB_ITERS=2
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o p -- python3 $root/tools/kbench.py rb_fwd > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o p -- python3 $root/tools/kbench.py rb_fwd > $out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_ANY --kernel-trace -d $out/pmc_sq -o p -- python3 $root/tools/kbench.py rb_fwd > $out/pmc_sq.log 2>&1
ls $out
