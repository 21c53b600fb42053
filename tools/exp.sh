cd $GRAFT_REPO_ROOT
python -m pytest tests/test_conv_gpu.py -m gpu -q -p no:cacheprovider -k "s2 or s221 or up_ or strided" 2>&1 | tail -3
for v in 1 0; do echo "== NNDET_IGEMM_LDSV2=$v"; NNDET_IGEMM_LDSV2=$v python tools/conv_microbench.py e1_32to64_s2 e2_64to128_s2 e3_128to256_s2 2>&1 | grep -v "Warn\|amdgpu"; done
