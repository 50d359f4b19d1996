#!/bin/bash
# round 3: rocprofv3 --pmc passes (SQ x2, FETCH_SIZE, WRITE_SIZE -- each its own run) of the residual-block backward / forward kernels
# and of the strided-layer backward at the bench shapes, per width.  Summaries -> gpurun_out/pmc_<tag>/summary.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export KB_N=3 KB_D=1,2,3 KB_WHAT=fwd,bwd,stride
for C in 32 16 8 4; do
  KB_C=$C bash tools/pmc_level.sh r03_C$C > /dev/null 2>&1
  echo "== C=$C"; grep -c "^k_" gpurun_out/pmc_r03_C$C/summary.txt
done
KB_C=32,16 KB_WHAT=bwdf bash tools/pmc_level.sh r03_fused > /dev/null 2>&1
KB_ITERS=3 PMC_PY=tools/kbench.py PMC_ARGS=cqt bash tools/pmc_level.sh r03_cqt > /dev/null 2>&1
KB_C=32,16,8,4 KB_D=1 KB_WHAT=stridefwd,edge bash tools/pmc_level.sh r03_misc > /dev/null 2>&1
ls gpurun_out/pmc_r03_*/summary.txt
