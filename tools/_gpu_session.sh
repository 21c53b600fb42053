cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/gpu_round.sh prof > gpurun_out/prof_stage.txt 2>&1; grep "k_pack" gpurun_out/kernel_stats.txt | head; grep "k_pack\|k_stem_fwd3<unsigned short, 1>" gpurun_out/timeline.txt | head
