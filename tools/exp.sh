cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_abi.py tests/test_parity_full_gpu.py -q -p no:cacheprovider --tb=short 2>&1 | tail -12
for v in 1 0 1 0; do echo -n "HEAD_GATHER=$v "; NNDET_HEAD_GATHER=$v python bench.py --steps 60 --warmup 10 --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"])"; done
