#!/bin/bash
# A/B of two builds of the library in ONE GPU session: nndetection_amd/csrc/libnndet_amd_prev.so (built from the commit to compare against) vs the
# current one. usage: tools/ab_lib.sh [pairs] [kernel-name pattern for the per-kernel comparison]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
P=$PWD/nndetection_amd/csrc/libnndet_amd_prev.so
PAIRS=${1:-4}; PAT=${2:-}
: > $O/ab_lib.txt
if [ -n "$PAT" ]; then
  for lib in prev cur; do
    rm -rf $O/prof
    L=$PWD/nndetection_amd/csrc/libnndet_amd.so; [ $lib = prev ] && L=$P
    (cd /tmp && NNDET_AMD_LIB=$L timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-extras > /dev/null 2>&1)
    db=$(find $O/prof -name "*_results.db" | head -1)
    echo "== lib=$lib (7 steps under rocprofv3 --kernel-trace)" | tee -a $O/ab_lib.txt
    [ -n "$db" ] && python tools/rocpd_stats.py "$db" 80 | grep -i "$PAT\|total kernel time" | cut -c1-150 | tee -a $O/ab_lib.txt
    rm -rf $O/prof
  done
fi
run() { env "$@" timeout 600 python bench.py --steps 80 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
run NNDET_AMD_LIB=$P > /dev/null
for r in $(seq $PAIRS); do
  echo "prev $(run NNDET_AMD_LIB=$P)" | tee -a $O/ab_lib.txt
  echo "cur  $(run X=1)" | tee -a $O/ab_lib.txt
done
