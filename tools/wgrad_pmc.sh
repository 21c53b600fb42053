#!/bin/bash
# SQ / LDS / GRBM counters of the weight-gradient kernels alone (tools/conv_microbench.py, wgrad only): where do the wave cycles go?
#   tools/wgrad_pmc.sh <out.txt> [shape ...]        (run on the GPU box; two --pmc passes, kernel trace only)
out=${1:-gpurun_out/wgrad_pmc.txt}; shift
shapes=${@:-e0_32x32_full e1_64x64 e1_32to64_s2}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/wgpmc; mkdir -p /tmp/wgpmc
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  MICRO_ORDER=wgrad MICRO_ITERS=3 rocprofv3 --pmc $P -d /tmp/wgpmc/p$i -- python /root/repo/tools/conv_microbench.py $shapes > /tmp/wgpmc/log$i.txt 2>&1
done
python /root/repo/tools/rocpd_pmc.py $(find /tmp/wgpmc -name "*_results.db") 2>&1 | grep -A20 "k_wgrad" > /root/repo/$out
tail -3 /tmp/wgpmc/log1.txt >> /root/repo/$out
