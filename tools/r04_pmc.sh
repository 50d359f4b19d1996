#!/bin/bash
# round 4: rocprofv3 --pmc passes (SQ x2, FETCH_SIZE, WRITE_SIZE -- each its own run) of the wide residual-block kernels at the bench shape:
# forward + default backward dispatch + the one-pass strip kernel forced at both widths.  Summaries -> gpurun_out/pmc_r04_*/summary.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
export KB_N=3 KB_D=1,2,3
KB_C=32 KB_WHAT=fwd,bwd bash tools/pmc_level.sh r04_C32 > /dev/null 2>&1
KB_C=32,16 KB_WHAT=bwd1 bash tools/pmc_level.sh r04_bwds > /dev/null 2>&1
ls gpurun_out/pmc_r04_*/summary.txt
