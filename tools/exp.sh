cd $GRAFT_REPO_ROOT
timeout 300 python tools/sync_audit.py 2>&1 | grep -v "amdgpu\|Warn" | tail -4
NNDET_SYNCFREE_LOSS=0 timeout 300 python tools/sync_audit.py 2>&1 | grep -v "amdgpu\|Warn" | tail -1
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_parity_full_gpu.py tests/test_sampler_gpu.py -m gpu -q -p no:cacheprovider --tb=short 2>&1 | tail -12
