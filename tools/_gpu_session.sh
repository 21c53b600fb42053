cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=long -p no:cacheprovider -k "persistent_strided" 2>&1 | grep -v "^$" | tail -40
