#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
P=$PWD/nndetection_amd/csrc/libnndet_amd_prev.so
for lib in prev cur; do
  rm -rf $O/prof
  L=$PWD/nndetection_amd/csrc/libnndet_amd.so; [ $lib = prev ] && L=$P
  (cd /tmp && NNDET_AMD_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-extras > /dev/null 2>&1)
  db=$(find $O/prof -name "*_results.db" | head -1)
  echo "== lib=$lib" | tee -a $O/ab_norm_shfl.txt
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" 60 | grep -i "k_norm_bwd_reduce\|total kernel time" | cut -c1-150 | tee -a $O/ab_norm_shfl.txt
  rm -rf $O/prof
done
run() { env "$@" timeout 600 python bench.py --steps 80 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
run NNDET_AMD_LIB=$P > /dev/null
for r in 1 2 3 4 5; do
  echo "prev $(run NNDET_AMD_LIB=$P)" | tee -a $O/ab_norm_shfl.txt
  echo "cur  $(run X=1)" | tee -a $O/ab_norm_shfl.txt
done
