#!/bin/bash
# extra rocprofv3 --pmc passes (issue mix and memory-pipe stalls); usage: tools/pmc_run2.sh <tag> <kbench args...>
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_ANY --kernel-trace -d $root/gpurun_out/pmc_${tag}_mix -o p -- python3 $root/tools/kbench.py "$@" > $root/gpurun_out/pmc_${tag}_mix.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES --kernel-trace -d $root/gpurun_out/pmc_${tag}_cnt -o p -- python3 $root/tools/kbench.py "$@" > $root/gpurun_out/pmc_${tag}_cnt.log 2>&1
# (TA_* / TCP_* counter passes hang under rocprofv3 on this pool: left out)
