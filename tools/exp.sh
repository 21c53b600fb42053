cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -5
for v in 1 0 1 0; do echo -n "IG3R=$v "; NNDET_IG3R=$v python bench.py --steps 40 --warmup 8 --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"])"; done
