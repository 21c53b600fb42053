#!/bin/bash
# Whole-step PMC survey per kernel: LDS bank conflicts and issue utilisation (which kernels are issue- or conflict-bound?)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
rm -rf $O/pmc_step
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  tag=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && timeout 600 rocprofv3 --pmc $pass -d $OLDPWD/$O/pmc_step/$tag -- python $OLDPWD/bench.py --steps 3 --warmup 2 --no-extras > /dev/null 2>&1)
done
python tools/rocpd_pmc.py $(find $O/pmc_step -name "*_results.db") > $O/pmc_step_summary.txt 2>&1
rm -rf $O/pmc_step
wc -l $O/pmc_step_summary.txt
