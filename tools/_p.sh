python -m pytest tests/test_gpu_wide_bf16.py -x -q -k "edge or channels_last" 2>&1 | tail -2
mkdir -p gpurun_out/e3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/e3/prof -o t -- python3 /root/repo/bench.py --timed-only --steps 6 --warmup 2 > /root/repo/gpurun_out/e3/bench.json 2> /root/repo/gpurun_out/e3/bench.err
tail -1 /root/repo/gpurun_out/e3/bench.json | cut -c1-200
