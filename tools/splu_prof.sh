timeout 900 python -m pytest tests/test_kron_gpu.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d['kron'],indent=1)); print(json.dumps(d['splu'],indent=1))"
