#!/bin/bash
# Scratch entry point of a gpurun call (`gpurun -- 'bash tools/_gpu_session.sh'`): edited per session. As committed: the profile set of HEAD.
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out
echo "== full GPU suite"; timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -4 | cut -c1-300 | tee $O/t_full.txt
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench (driver flags)"; timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.txt 2>$O/bench_err.txt; tail -c 200 $O/bench_driver_flags.txt
echo "== phase times"; timeout 300 python tools/phase_times.py > $O/phase_times.txt 2>&1; tail -3 $O/phase_times.txt
echo "== prof"; tools/gpu_round.sh prof > $O/prof_stdout.txt 2>&1; head -6 $O/kernel_stats.txt | cut -c1-150; head -3 $O/timeline.txt
echo "== step traffic"; tools/gpu_round.sh steptraffic > $O/steptraffic_stdout.txt 2>&1; tail -20 $O/steptraffic_stdout.txt | head -16 | cut -c1-120
