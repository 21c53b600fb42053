#!/bin/bash
# Process-level search over NNDET_STREAM_PADS (which side streams share a hardware queue): one bench run per configuration.
cd "$(dirname "$0")/.."
run() { echo -n "$1  "; NNDET_STREAM_PADS="$1" timeout 300 python bench.py --steps 40 --warmup 12 --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for cfg in "$@"; do run "$cfg"; done
