mkdir -p gpurun_out/l3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/l3/prof -o t -- python3 /root/repo/bench.py --timed-only --steps 6 --warmup 2 > /root/repo/gpurun_out/l3/bench.json 2> /root/repo/gpurun_out/l3/bench.err
tail -1 /root/repo/gpurun_out/l3/bench.json | cut -c1-200
