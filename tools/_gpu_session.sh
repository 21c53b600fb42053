cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_pyramid_gpu.py tests/test_norm_builds_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "norm or block or items or trunk or builds" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_parity_full_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "inside_the_step" 2>&1 | tail -2
rm -f gpurun_out/ablib.txt; bash tools/gpu_round.sh ablib 2>&1 | tail -4
