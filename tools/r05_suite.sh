#!/bin/bash
# the whole GPU selection as the driver runs it (wall-clock limit), then the slow-marked re-runs of the opt-in / A-B paths, then smoke()
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
/usr/bin/time -v python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/r05_suite.log 2>&1
tail -25 gpurun_out/r05_suite.log
TT_RUN_SLOW=1 python -m pytest tests/test_gpu_wide_bf16.py -q -m gpu -k "separate_kernel or recompute_path" > gpurun_out/r05_suite_slow.log 2>&1
tail -3 gpurun_out/r05_suite_slow.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; tail -5 gpurun_out/r05_smoke.log
