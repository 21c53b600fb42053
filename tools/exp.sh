cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --tb=short -k "norm or tiny or deferred or materialize" 2>&1 | tail -4
for i in 1 2; do python bench.py --steps 60 --warmup 10 --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"])"; done
