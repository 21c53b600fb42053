#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
echo "== suite"
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/t_suite.txt 2>&1; tail -8 $O/t_suite.txt; grep -n "^E  " $O/t_suite.txt | cut -c1-300 | head -20
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== gioupmc / nmspmc"
tools/gpu_round.sh gioupmc nmspmc > $O/pmc_stdout.txt 2>&1; grep "derived\|^k_\|^void k_" $O/gioupmc_summary.txt | head -30; grep "derived\|^k_nms" $O/nmspmc_summary.txt | head -40
echo "== torch baseline fair"
NNDET_TORCH_BASELINE_MODE=fair timeout 1500 python bench.py --torch-baseline-child --plan luna160 --batch 4 > $O/torch_fair.txt 2>&1; tail -1 $O/torch_fair.txt
