#!/bin/bash
# Shader clock and socket power sampled with rocm-smi while bench.py runs (round 4: is the training step power / clock bound?)
#   tools/clock_probe.sh [out.txt] [bench args...]
out=${1:-gpurun_out/clock_probe.txt}; shift
python bench.py --steps 2500 --warmup 15 --no-extras "$@" > /tmp/cp_bench.txt 2>&1 &
pid=$!
: > $out
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed 's/.*sclk clock level: [0-9S]*: //; s/.*Power (W): /W /' | tr '\n' ' ' >> $out; echo >> $out
  sleep 0.3
done
grep -o '"ms_per_step": [0-9.]*' /tmp/cp_bench.txt >> $out
