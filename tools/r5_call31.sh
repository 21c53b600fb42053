#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "s2 or strided" 2>&1 | tail -4 | cut -c1-300
P=$PWD/nndetection_amd/csrc/libnndet_amd_prev.so
echo "== conv_microbench e1_32to64_s2 (forward = k_ig3s): prev lib, then current" | tee $O/ig3s_swz.txt
NNDET_AMD_LIB=$P timeout 300 python tools/conv_microbench.py e1_32to64_s2 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O/ig3s_swz.txt
timeout 300 python tools/conv_microbench.py e1_32to64_s2 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O/ig3s_swz.txt
rm -rf $O/pmc_ig3s
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY -d $OLDPWD/$O/pmc_ig3s/a -- python $OLDPWD/tools/conv_microbench.py e1_32to64_s2 > /dev/null 2>&1)
python tools/rocpd_pmc.py $(find $O/pmc_ig3s -name "*_results.db") 2>&1 | grep -A8 "k_ig3s" | head -12 | tee -a $O/ig3s_swz.txt
rm -rf $O/pmc_ig3s
timeout 900 python -m pytest tests/test_parity_full_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "luna160 or toy64 or tiny" 2>&1 | tail -3 | cut -c1-300
