#!/bin/bash
# Round 5 profile set at HEAD: bench line (default flags), kernel statistics + one-step timeline, whole-step PMC traffic, phase times.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
echo "== failing test rerun"
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "optimizer_steps_track or foreign_fused" 2>&1 | tail -3
echo "== bench (default flags)"
timeout 900 python bench.py > $O/bench.txt 2>&1; tail -1 $O/bench.txt | cut -c1-400
echo "== phase times"
timeout 300 python tools/phase_times.py > $O/phase_times.txt 2>&1; tail -12 $O/phase_times.txt
echo "== prof"
tools/gpu_round.sh prof > $O/prof_stdout.txt 2>&1; head -28 $O/kernel_stats.txt | cut -c1-150; head -3 $O/timeline.txt
echo "== step traffic"
tools/gpu_round.sh steptraffic > $O/steptraffic_stdout.txt 2>&1; tail -25 $O/steptraffic_stdout.txt | cut -c1-200
