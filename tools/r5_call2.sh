#!/bin/bash
# Round 5, GPU call 2: switch sweep in one session (alternating with the default), forced world-1 RCCL path, GPU suite.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
run() { env "$@" timeout 600 python bench.py --steps 60 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
echo "== switch sweep (each: default, switch, default, switch)" | tee $O/ab_switches.txt
for sw in NNDET_DEFER_NORM=1 NNDET_WGRAD_PRIO=0 NNDET_STEM_BWD_MAIN=0 NNDET_WGRAD3D_WGS=192 NNDET_WGRAD3S_WGS=192 NNDET_HEAD_STREAMS=0 NNDET_OVERLAP_AUX=0; do
  for rep in 1 2; do
    echo "default        $(run X=1)" | tee -a $O/ab_switches.txt
    echo "$sw  $(run $sw)" | tee -a $O/ab_switches.txt
  done
done
echo "== forced world-1 RCCL path" | tee $O/force_dist.txt
for v in 1 0 1 0; do NNDET_BENCH_FORCE_DIST=1 NNDET_DDP_INPLACE=$v timeout 600 python bench.py --steps 60 --warmup 15 --no-extras > $O/fd_$v.txt 2>&1
  echo "INPLACE=$v $(grep -o '"ms_per_step": [0-9.]*' $O/fd_$v.txt | head -1) $(grep -o '"ddp": {[^}]*}' $O/fd_$v.txt | head -1)" | tee -a $O/force_dist.txt; done
echo "plain $(run X=1)" | tee -a $O/force_dist.txt
echo "== suite"
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/t_suite.txt 2>&1; tail -25 $O/t_suite.txt
