cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
NNDET_AMD_LIB=$PWD/nndetection_amd/csrc/libnndet_amd_timing.so MICRO_ITERS=2 MICRO_ORDER=wgrad timeout 300 python tools/conv_microbench.py e0_32x32_full e1_64x64 2>&1 | grep -v amdgpu.ids | tail -60 | tee gpurun_out/wg3e_timing.txt
