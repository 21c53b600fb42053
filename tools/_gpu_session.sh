cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for i in 1 2 3; do timeout 900 python -m pytest tests/test_parity_full_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "luna160_bf16_end_to_end" > gpurun_out/t_bf16_$i.txt 2>&1; tail -3 gpurun_out/t_bf16_$i.txt; grep -i "cos\|norm" gpurun_out/parity_bf16.txt | tail -6; done
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_pyramid_gpu.py tests/test_plugin_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/t_model.txt 2>&1; tail -5 gpurun_out/t_model.txt
