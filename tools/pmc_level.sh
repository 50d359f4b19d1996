#!/bin/bash
# rocprofv3 --pmc passes over tools/kb_level.py (env KB_* selects the kernels); usage: tools/pmc_level.sh <tag>
# Each pass is its own run (counters only with --kernel-trace); summary text written next to the databases.
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export KB_N=${KB_N:-4}
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $out/sq1 -o p -- python3 $root/tools/kb_level.py > $out/sq1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace -d $out/sq2 -o p -- python3 $root/tools/kb_level.py > $out/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/fetch -o p -- python3 $root/tools/kb_level.py > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/write -o p -- python3 $root/tools/kb_level.py > $out/write.log 2>&1
cd $root
python3 tools/pmc_summary.py k_ $(find $out -name '*_results.db') > $out/summary.txt 2>&1
cat $out/sq1.log
cat $out/summary.txt
