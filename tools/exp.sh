cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do echo -n "HEAD_STREAMS=$v "; NNDET_HEAD_STREAMS=$v python bench.py --steps 60 --warmup 10 --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"])"; done
