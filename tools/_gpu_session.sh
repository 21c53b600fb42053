cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for v in 0 1 0 1; do NNDET_DGSP=$v timeout 300 python tools/phase_times.py 40 2>&1 | grep -v amdgpu.ids | tail -3 | sed "s/^/dgsp=$v /"; done | tee gpurun_out/phase_dgsp.txt
