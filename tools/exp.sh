cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -5
L=nndetection_amd/csrc/libnndet_amd.so
cp $L /tmp/new.so
for v in new old new old; do
  if [ $v = new ]; then cp /tmp/new.so $L; else cp build/libold.so $L; fi
  echo -n "$v: "; python bench.py --steps 60 --warmup 10 --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"])"
done
cp /tmp/new.so $L
MICRO_ITERS=20 python tools/conv_microbench.py 2>&1 | grep -v "Warn\|amdgpu"
