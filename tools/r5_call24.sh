#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "stem" 2>&1 | tail -4 | cut -c1-300
rm -rf $O/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -- python $OLDPWD/tools/stem_bwd_microbench.py 6 > $OLDPWD/$O/prof.txt 2>&1)
db=$(find $O/prof -name "*_results.db" | head -1)
echo "== swizzled $(grep 'dgamma' $O/prof.txt | cut -c1-200)" | tee $O/stem_bwd_swz.txt
[ -n "$db" ] && python tools/rocpd_stats.py "$db" 40 | grep -i "k_stem_bwd3" | cut -c1-150 | tee -a $O/stem_bwd_swz.txt
rm -rf $O/prof $O/pmc_stem
for pass in "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && timeout 300 rocprofv3 --pmc $pass -d $OLDPWD/$O/pmc_stem/$tag -- python $OLDPWD/tools/stem_bwd_microbench.py 3 > /dev/null 2>&1)
done
python tools/rocpd_pmc.py $(find $O/pmc_stem -name "*_results.db") 2>&1 | grep -A14 "k_stem_bwd3" | tee -a $O/stem_bwd_swz.txt
rm -rf $O/pmc_stem
run() { env "$@" timeout 600 python bench.py --steps 60 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
P=$PWD/nndetection_amd/csrc/libnndet_amd_prev.so
echo "== A/B (prev = the build before all k_stem_bwd3 changes)" | tee $O/ab_stem_swz.txt
run NNDET_AMD_LIB=$P > /dev/null
for r in 1 2 3; do
  echo "prev $(run NNDET_AMD_LIB=$P)" | tee -a $O/ab_stem_swz.txt
  echo "cur  $(run X=1)" | tee -a $O/ab_stem_swz.txt
done
