cd $GRAFT_REPO_ROOT
NNDET_IGEMM_SMALLWG=100000000 timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -p no:cacheprovider --tb=line -k "lib or norm_relu" 2>&1 | tail -8
for v in 0 100000; do echo "== SMALLWG=$v"; NNDET_IGEMM_SMALLWG=$v MICRO_ITERS=20 timeout 300 python tools/conv_microbench.py p2_128x128 p3_128x128 p4_128x128 e3_256x256 2>&1 | grep -v "Warn\|amdgpu" | sed 's/| wgrad.*//'; done
for v in 0 300 700 0 300 700; do echo -n "SMALLWG=$v "; NNDET_IGEMM_SMALLWG=$v python bench.py --steps 60 --warmup 10 --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"])"; done
