#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
echo "== suite"
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/t_suite.txt 2>&1; tail -6 $O/t_suite.txt; grep -n "^E  " $O/t_suite.txt | cut -c1-300 | head -20
echo "== bench (driver flags)"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.txt 2>&1; tail -1 $O/bench.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline_dominant'])[:1500]); print(json.dumps(d.get('step_roofline'))[:600])"
