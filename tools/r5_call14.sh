#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "norm_backward_sums" 2>&1 | tail -40 | cut -c1-250
for v in 3 4; do NNDET_DGS_NB_LATE=$v timeout 900 python -m pytest tests/test_parity_full_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "norm_backward_inside" 2>&1 | tail -8 | cut -c1-400; done
