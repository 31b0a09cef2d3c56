cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py --pods 200000 --steps 1 --warmup 1 --no-cpu-baseline --batch-problems 256 --batch-pods 20000 2>&1 | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(b['value'], b['ms_per_step'], b['batched'])"
timeout 900 python bench.py --pods 200000 --steps 1 --warmup 1 --no-cpu-baseline --batch-problems 512 --batch-pods 20000 2>&1 | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(b['value'], b['batched'])"
