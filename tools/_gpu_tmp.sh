cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for i in 1 2; do timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-torch-baseline --no-routes 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['timed_blocks']['ms_per_step'], d['chip_probe']['k_ig3r_forward_alone_ms'])"; done
