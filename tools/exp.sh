cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --tb=short -k "ddp or streams" 2>&1 | tail -4
for i in 1 2; do
python bench.py --steps 60 --warmup 10 --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d[\"value\"], d[\"ms_per_step\"])"
NNDET_BENCH_FORCE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 60 --warmup 10 --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('force_dist', d[\"value\"], d[\"ms_per_step\"])"
done
