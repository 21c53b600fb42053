cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_conv_gpu.py tests/test_pyramid_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -3
MICRO_ITERS=20 MICRO_ORDER=wgrad,wgrad timeout 300 python tools/conv_microbench.py e0_32x32_full e1_64x64 p2_128x128 e3_256x256 e4_320x320 e1_32to64_s2 lat_p1_1x1 2>&1 | grep -v amdgpu.ids
NNDET_AMD_LIB=$PWD/nndetection_amd/csrc/libnndet_amd_prev.so MICRO_ITERS=20 MICRO_ORDER=wgrad,wgrad timeout 300 python tools/conv_microbench.py e0_32x32_full e1_64x64 p2_128x128 e3_256x256 e4_320x320 e1_32to64_s2 lat_p1_1x1 2>&1 | grep -v amdgpu.ids | sed "s/^/prev /"
rm -f gpurun_out/ablib.txt; bash tools/gpu_round.sh ablib 2>&1 | tail -4
