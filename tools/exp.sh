cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q -p no:cacheprovider -k "ig3r" --tb=short 2>&1 | tail -30
for v in 0 1; do echo "== NNDET_IG3R=$v"; NNDET_IG3R=$v timeout 300 python tools/conv_microbench.py e0_32x32_full 2>&1 | grep -v "Warn\|amdgpu"; done
