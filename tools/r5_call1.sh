#!/bin/bash
# Round 5, GPU call 1: A/B of the segmentation-branch order and of the norm-reduction kernels, bench, micro-benchmarks, suite, profile.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
run() { env "$@" timeout 600 python bench.py --steps 60 --warmup 15 --no-extras 2>&1 | grep -o '"ms_per_step": [0-9.]*' | head -1; }
echo "== A/B NNDET_SEG_FIRST (cur lib)" | tee $O/ab_segfirst.txt
for v in 1 0 1 0; do echo "SEG_FIRST=$v $(run NNDET_SEG_FIRST=$v)" | tee -a $O/ab_segfirst.txt; done
echo "== A/B library cur vs prev (norm reductions), SEG_FIRST=1" | tee $O/ab_normlib.txt
for v in cur prev cur prev; do lib=$PWD/nndetection_amd/csrc/libnndet_amd.so; [ $v = prev ] && lib=$PWD/nndetection_amd/csrc/libnndet_amd_prev.so
  echo "lib=$v $(run NNDET_AMD_LIB=$lib)" | tee -a $O/ab_normlib.txt; done
echo "== norm microbench cur / prev" | tee $O/norm_micro.txt
timeout 300 python tools/norm_microbench.py 2>&1 | tee -a $O/norm_micro.txt
NNDET_AMD_LIB=$PWD/nndetection_amd/csrc/libnndet_amd_prev.so timeout 300 python tools/norm_microbench.py 2>&1 | sed 's/^/prev: /' | tee -a $O/norm_micro.txt
echo "== forced world-1 RCCL path (in place)" | tee $O/force_dist.txt
for v in 1 0; do echo "INPLACE=$v $(NNDET_BENCH_FORCE_DIST=1 NNDET_DDP_INPLACE=$v timeout 600 python bench.py --steps 60 --warmup 15 --no-extras 2>&1 | tail -1 | cut -c1-1500)" | tee -a $O/force_dist.txt; done
echo "plain $(run X=1)" | tee -a $O/force_dist.txt
echo "== bench (default flags)"
timeout 900 python bench.py > $O/bench.txt 2>&1; tail -1 $O/bench.txt | cut -c1-600
echo "== box microbench"
timeout 600 python tools/box_microbench.py > $O/box_micro.txt 2>&1; cat $O/box_micro.txt
echo "== suite"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $O/t_suite.txt 2>&1; tail -15 $O/t_suite.txt
echo "== prof"
tools/gpu_round.sh prof > $O/prof_stdout.txt 2>&1; head -30 $O/kernel_stats.txt | cut -c1-160
