#!/bin/bash
# separate rocprofv3 --pmc passes (kernel-trace only) over one kbench invocation; usage: tools/pmc_run.sh <tag> <kbench args...>
# env (KB_C, TTRAP_PRECISION ...) is inherited; output under gpurun_out/pmc_<tag>_{fetch,write,sq}
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"; do
  set -- $pass "__SEP__" "$@"
  name=$1; shift; ctrs=()
  while [ "$1" != "__SEP__" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctrs[@]}" --kernel-trace -d $root/gpurun_out/pmc_${tag}_$name -o p -- python3 $root/tools/kbench.py "$@" > $root/gpurun_out/pmc_${tag}_$name.log 2>&1
done
