cd $GRAFT_REPO_ROOT
for m in e0_32x32_full lat_p0_1x1 up_p1_64to32 e1_32to64_s2; do
  MICRO=$m bash tools/gpu_round.sh pmc > /dev/null 2>&1
  cp gpurun_out/pmc_summary.txt gpurun_out/pmc_$m.txt
done
ls gpurun_out
