cd $GRAFT_REPO_ROOT
python -m pytest tests/test_boxes_gpu.py tests/test_wbc_gpu.py tests/test_postprocess_gpu.py -m gpu -q -p no:cacheprovider -k "nms or wbc or postprocess" 2>&1 | tail -3
python tools/box_microbench.py --no-cpu 2>&1 | grep -i "nms"
