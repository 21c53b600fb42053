cd $GRAFT_REPO_ROOT
python -m pytest tests/test_conv_gpu.py -m gpu -q -p no:cacheprovider -k "ig3 or c32_k3 or c64_k3 or c128" 2>&1 | tail -2
for v in 1 2; do echo "== NNDET_WGRAD3_QD=$v"; NNDET_WGRAD3_QD=$v python tools/conv_microbench.py e0_32x32_full e1_64x64 p2_128x128 head_reg_out 2>&1 | grep -v "Warn\|amdgpu" | sed 's/.*| wgrad/wgrad/'; done
