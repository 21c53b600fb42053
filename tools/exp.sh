cd $GRAFT_REPO_ROOT
python -m pytest tests/test_conv_gpu.py -m gpu -q -p no:cacheprovider -k "lateral or up_ or seg_out or transposed" 2>&1 | tail -8
for v in 0 1; do echo "== NNDET_PW=$v"; NNDET_PW=$v python tools/conv_microbench.py lat_p0_1x1 lat_p1_1x1 up_p1_64to32 up_p2_128to64 2>&1 | grep -v "Warn\|amdgpu.ids"; done
