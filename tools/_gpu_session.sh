cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -f gpurun_out/ablib.txt; bash tools/gpu_round.sh ablib 2>&1 | tail -4;  bash tools/gpu_round.sh ablib 2>&1 | tail -4
