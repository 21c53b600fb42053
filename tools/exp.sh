cd $GRAFT_REPO_ROOT
MICRO=p2_128x128 bash tools/gpu_round.sh pmc > /dev/null 2>&1
grep -A16 "k_ig3" gpurun_out/pmc_summary.txt | head -40
