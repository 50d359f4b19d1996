#!/bin/bash
# the whole GPU selection as the driver runs it (wall-clock limit), then the slow-marked re-runs of the opt-in / A-B paths, then smoke()
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
t0=$(date +%s)
python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/r05_suite.log 2>&1
echo "suite wall seconds: $(( $(date +%s) - t0 ))" >> gpurun_out/r05_suite.log
tail -25 gpurun_out/r05_suite.log
