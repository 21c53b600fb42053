#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "stem" 2>&1 | tail -4 | cut -c1-300
rm -rf $O/prof
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -- python $OLDPWD/tools/stem_bwd_microbench.py 6 > $OLDPWD/$O/prof.txt 2>&1)
db=$(find $O/prof -name "*_results.db" | head -1)
echo "== no valid select $(grep 'dgamma' $O/prof.txt | cut -c1-200)" | tee $O/stem_bwd_v3.txt
[ -n "$db" ] && python tools/rocpd_stats.py "$db" 40 | grep -i "k_stem_bwd3" | cut -c1-150 | tee -a $O/stem_bwd_v3.txt
rm -rf $O/prof
timeout 900 python -m pytest tests/test_parity_full_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "luna160 or toy64 or tiny" 2>&1 | tail -4 | cut -c1-300
