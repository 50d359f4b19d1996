#!/bin/bash
# round 5: rocprofv3 --pmc passes over tools/kb_stride.py (strided / transposed layers at the bench shapes, forward, backward, backward pregated)
cd "$(dirname "$0")/.."
export KB_N=3
PMC_PY=tools/kb_stride.py bash tools/pmc_level.sh r05_stride > /dev/null 2>&1
grep -A22 "k_w4<32, false, true, true, true>\|k_w4<32, true, true, true, false>\|k_w4<32, false, true, false" gpurun_out/pmc_r05_stride/summary.txt | head -80
cat gpurun_out/pmc_r05_stride/timings_under_profiler.txt | head -40
