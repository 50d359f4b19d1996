python -m pytest tests/test_gpu_wide_bf16.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
