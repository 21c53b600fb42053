#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "norm_backward_sums and prefetch" 2>&1 | tail -6 | cut -c1-400
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "norm_backward_sums and late_loads" 2>&1 | tail -6 | cut -c1-400
timeout 900 python -m pytest tests/test_parity_full_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "norm_backward_inside" 2>&1 | tail -6 | cut -c1-400
run() { env "$@" timeout 600 python bench.py --steps 60 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
echo "== A/B NNDET_NORM_RED_FUSE (0 = separate k_norm_bwd_reduce; 1 = sums in k_dgs's epilogue; late = residual / y loaded after the MFMA loop)" | tee $O/ab_norm_red_fuse.txt
for r in 1 2 3; do
  echo "FUSE=0      $(run NNDET_NORM_RED_FUSE=0)" | tee -a $O/ab_norm_red_fuse.txt
  echo "FUSE=1      $(run NNDET_NORM_RED_FUSE=1)" | tee -a $O/ab_norm_red_fuse.txt
  echo "FUSE=1 late $(run NNDET_NORM_RED_FUSE=1 NNDET_DGS_NB_LATE=1)" | tee -a $O/ab_norm_red_fuse.txt
done
