#!/bin/bash
# round 5, first GPU session: the new tests first (fast feedback), then the whole GPU suite with durations, the bench line, the fp16 study
mkdir -p gpurun_out
python -m pytest tests/test_gpu_cqt.py tests/test_gpu_distributed.py -q -m gpu -k "generic or other_block or rccl_single or conventions or inverse or round_trip" > gpurun_out/r05_t1.log 2>&1
python -m pytest tests/test_gpu_model.py tests/test_gpu_wide_bf16.py -q -m gpu -k "loss_scale or overflow or frozen or fused_consistency or fp16_hidden" > gpurun_out/r05_t2.log 2>&1
python -m pytest tests -q -m gpu --durations=80 > gpurun_out/r05_full.log 2>&1
python bench.py > gpurun_out/r05_bench0.json 2> gpurun_out/r05_bench0.err
DIAG_B=64 python tools/diag/fp16_vs_bf16.py > gpurun_out/r05_fp16.txt 2>&1
tail -3 gpurun_out/r05_t1.log gpurun_out/r05_t2.log gpurun_out/r05_full.log
