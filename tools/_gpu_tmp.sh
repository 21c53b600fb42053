cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_pyramid_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_parity_full_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "b4 or luna160" 2>&1 | tail -2
bash tools/gpu_round.sh prof > gpurun_out/prof_stage.txt 2>&1; grep "k_ho_" gpurun_out/timeline.txt | cut -c1-110
rm -f gpurun_out/ablib.txt; bash tools/gpu_round.sh ablib 2>&1 | tail -4
