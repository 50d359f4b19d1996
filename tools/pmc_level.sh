#!/bin/bash
# rocprofv3 --pmc passes over tools/kb_level.py (env KB_* selects the kernels) or PMC_PY / PMC_ARGS; usage: tools/pmc_level.sh <tag>
# Each pass is its own run (counters only with --kernel-trace).  The databases stay in /tmp on the box (they exceed what gpurun
# copies back); the per-kernel averages are written to gpurun_out/pmc_<tag>/summary.txt.
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
db=/tmp/pmc_db_$tag
mkdir -p $out $db
cd /tmp && export TMPDIR=/tmp
export KB_N=${KB_N:-4}
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $db/sq1 -o p -- python3 $root/${PMC_PY:-tools/kb_level.py} ${PMC_ARGS:-} > $out/sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d $db/sq2 -o p -- python3 $root/${PMC_PY:-tools/kb_level.py} ${PMC_ARGS:-} > $out/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $db/fetch -o p -- python3 $root/${PMC_PY:-tools/kb_level.py} ${PMC_ARGS:-} > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $db/write -o p -- python3 $root/${PMC_PY:-tools/kb_level.py} ${PMC_ARGS:-} > $out/write.log 2>&1
cd $root
python3 tools/pmc_summary.py k_ $(find $db -name '*_results.db') > $out/summary.txt 2>&1
grep -h "ms " $out/fetch.log > $out/timings_under_profiler.txt
rm -rf $db
