cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/fpmc; mkdir -p /tmp/fpmc
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES"
i=0
for P in "$P1" "$P2"; do i=$((i+1)); MICRO_ORDER=fwd,dgrad MICRO_ITERS=3 rocprofv3 --pmc $P -d /tmp/fpmc/p$i -- python /root/repo/tools/conv_microbench.py e1_32to64_s2 e2_64to128_s2 > /tmp/fpmc/log$i.txt 2>&1; done
python /root/repo/tools/rocpd_pmc.py $(find /tmp/fpmc -name "*_results.db") 2>&1 | grep -A17 "k_igemm\|k_dgs" | head -80
tail -2 /tmp/fpmc/log1.txt
