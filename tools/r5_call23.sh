#!/bin/bash
# PMC counters of the two tail kernels changed in round 5 (k_dgs with the norm-backward epilogue, k_stem_bwd3), alone, separate passes.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
for mb in dgs_normred stem_bwd; do
  rm -rf $O/pmc_$mb
  for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $pass | cut -d' ' -f1)
    (cd /tmp && timeout 300 rocprofv3 --pmc $pass -d $OLDPWD/$O/pmc_$mb/$tag -- python $OLDPWD/tools/${mb}_microbench.py 3 > $OLDPWD/$O/pmc_${mb}_$tag.txt 2>&1)
  done
  python tools/rocpd_pmc.py $(find $O/pmc_$mb -name "*_results.db") > $O/pmc_${mb}_summary.txt 2>&1
  rm -rf $O/pmc_$mb
done
grep -A26 "k_dgs" $O/pmc_dgs_normred_summary.txt | head -40
grep -A26 "k_stem_bwd3" $O/pmc_stem_bwd_summary.txt | head -40
