#!/bin/bash
# rocprofv3 --pmc passes (kernel-trace only) over the narrow-level residual block kernels; usage: tools/pmc_narrow.sh <tag> <C>
tag=$1; C=$2
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export KB_C=$C KB_ITERS=2
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $root/gpurun_out/pmc_${tag}_a -o p -- python3 $root/tools/kbench.py rb_fwd rb_bwd > $root/gpurun_out/pmc_${tag}_a.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM --kernel-trace -d $root/gpurun_out/pmc_${tag}_b -o p -- python3 $root/tools/kbench.py rb_fwd rb_bwd > $root/gpurun_out/pmc_${tag}_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $root/gpurun_out/pmc_${tag}_f -o p -- python3 $root/tools/kbench.py rb_fwd rb_bwd > $root/gpurun_out/pmc_${tag}_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $root/gpurun_out/pmc_${tag}_w -o p -- python3 $root/tools/kbench.py rb_fwd rb_bwd > $root/gpurun_out/pmc_${tag}_w.log 2>&1
