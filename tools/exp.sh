cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_parity_full_gpu.py -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -6
for v in 1 0 1 0; do echo -n "FUSE_GRAD_ACC=$v "; NNDET_FUSE_GRAD_ACC=$v python bench.py --steps 60 --warmup 10 --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"])"; done
