#!/bin/bash
# One GPU-box session: probes, parity tests (each file in its own process), smoke, bench, rocprof. Everything is
# written under gpurun_out/ (merged back by gpurun). Usage: tools/gpu_round.sh [stages...]  (default: all)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
STAGES=${@:-probe boxes conv model smoke bench prof}
echo "== host: $(nproc) cores; $(rocminfo 2>/dev/null | grep -m1 -E 'gfx9[0-9a-f]+' )" | tee gpurun_out/host.txt
for s in $STAGES; do
  case $s in
    probe) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probe_mfma.hip -o /tmp/probe 2>/dev/null && timeout 60 /tmp/probe > gpurun_out/probe.txt 2>&1; head -4 gpurun_out/probe.txt;;
    boxes) timeout 900 python -m pytest tests/test_boxes_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_boxes.txt 2>&1; tail -25 gpurun_out/t_boxes.txt;;
    conv)  timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_conv.txt 2>&1; tail -40 gpurun_out/t_conv.txt;;
    model) timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_model.txt 2>&1; tail -30 gpurun_out/t_model.txt;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; tail -5 gpurun_out/smoke.txt;;
    bench) timeout 1200 python bench.py ${BENCH_ARGS:---steps 30 --warmup 5} > gpurun_out/bench.txt 2>&1; tail -5 gpurun_out/bench.txt;;
    prof)  rm -rf gpurun_out/prof; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof -- python $OLDPWD/bench.py --steps 3 --warmup 2 --no-extras > $OLDPWD/gpurun_out/prof.txt 2>&1); db=$(find gpurun_out/prof -name "*_results.db" | head -1); [ -n "$db" ] && python tools/rocpd_stats.py "$db" 5 --by-grid --timeline gpurun_out/timeline.txt > gpurun_out/kernel_stats.txt && head -22 gpurun_out/kernel_stats.txt | cut -c1-170; rm -rf gpurun_out/prof gpurun_out/prof_old; true;;
    micro) timeout 600 python tools/conv_microbench.py $MICRO > gpurun_out/micro.txt 2>&1; cat gpurun_out/micro.txt;;
    pmc)   rm -rf gpurun_out/pmc; for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_MFMA"; do
             tag=$(echo $pass | cut -d' ' -f1); (cd /tmp && timeout 300 rocprofv3 --pmc $pass -d $OLDPWD/gpurun_out/pmc/$tag -- python $OLDPWD/tools/conv_microbench.py ${MICRO:-e0_32x32_full} > $OLDPWD/gpurun_out/pmc_$tag.txt 2>&1); done
           python tools/rocpd_pmc.py $(find gpurun_out/pmc -name "*_results.db") > gpurun_out/pmc_summary.txt 2>&1; head -60 gpurun_out/pmc_summary.txt; rm -rf gpurun_out/pmc;;
    nmspmc) rm -rf gpurun_out/nmspmc; for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES"; do
             tag=$(echo $pass | cut -d' ' -f1); (cd /tmp && timeout 300 rocprofv3 --pmc $pass -d $OLDPWD/gpurun_out/nmspmc/$tag -- python $OLDPWD/tools/nms_microbench.py 10000 3 > $OLDPWD/gpurun_out/nmspmc_$tag.txt 2>&1); done
           python tools/rocpd_pmc.py $(find gpurun_out/nmspmc -name "*_results.db") > gpurun_out/nmspmc_summary.txt 2>&1; grep -A12 "k_nms" gpurun_out/nmspmc_summary.txt | head -70; rm -rf gpurun_out/nmspmc;;
    gioupmc) rm -rf gpurun_out/gioupmc; for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_TRANS" "FETCH_SIZE WRITE_SIZE"; do
             tag=$(echo $pass | cut -d' ' -f1); (cd /tmp && timeout 300 rocprofv3 --pmc $pass -d $OLDPWD/gpurun_out/gioupmc/$tag -- python $OLDPWD/tools/giou_microbench.py 3 > $OLDPWD/gpurun_out/gioupmc_$tag.txt 2>&1); done
           python tools/rocpd_pmc.py $(find gpurun_out/gioupmc -name "*_results.db") > gpurun_out/gioupmc_summary.txt 2>&1; grep -A22 "k_pairwise" gpurun_out/gioupmc_summary.txt | head -60; rm -rf gpurun_out/gioupmc;;
    parity) timeout 1500 python -m pytest tests/test_parity_full_gpu.py tests/test_postprocess_gpu.py tests/test_targets_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_parity.txt 2>&1; tail -40 gpurun_out/t_parity.txt;;
    pyr)   timeout 900 python -m pytest tests/test_pyramid_gpu.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_pyr.txt 2>&1; tail -40 gpurun_out/t_pyr.txt;;
    ab)    for v in 1 0 1 0; do env "${AB_VAR:-NNDET_HEAD_ITEMS}=$v" timeout 600 python bench.py --steps 60 --warmup 15 --no-extras > gpurun_out/ab_items$v.txt 2>&1; echo "items=$v $(grep -o '"value": [0-9.]*' gpurun_out/ab_items$v.txt | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_items$v.txt | head -1)" | tee -a gpurun_out/ab.txt; done;;
    ablib) # A/B of two builds of the library in one session: nndetection_amd/csrc/libnndet_amd_prev.so (git archive <rev> + build.sh) vs the current one
           for v in cur prev cur prev; do lib=$PWD/nndetection_amd/csrc/libnndet_amd.so; [ $v = prev ] && lib=$PWD/nndetection_amd/csrc/libnndet_amd_prev.so
             NNDET_AMD_LIB=$lib timeout 600 python bench.py --steps 60 --warmup 15 --no-extras > gpurun_out/ablib_$v.txt 2>&1; echo "lib=$v $(grep -o '"value": [0-9.]*' gpurun_out/ablib_$v.txt | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ablib_$v.txt | head -1)" | tee -a gpurun_out/ablib.txt; done;;
    r4parity) timeout 1500 python -m pytest tests/test_parity_full_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "b4 or lidc192" > gpurun_out/t_r4parity.txt 2>&1; tail -30 gpurun_out/t_r4parity.txt;;
    r4model) timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "foreign_fused or optimizer_steps" > gpurun_out/t_r4model.txt 2>&1; tail -15 gpurun_out/t_r4model.txt;;
    steptraffic) timeout 600 python tools/step_traffic.py --steps 5 --warmup 3 > gpurun_out/step_traffic_stdout.txt 2>&1; cat gpurun_out/step_traffic_stdout.txt; head -20 gpurun_out/step_traffic.txt;;
    suite) timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/t_suite.txt 2>&1; tail -8 gpurun_out/t_suite.txt;;
  esac
done
