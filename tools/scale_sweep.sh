#!/bin/bash
# Gradient all-reduce tuning sweep for the first multi-GPU lease: runs `bench.py --gpus N` (one rank per GPU under
# torch.distributed.run, RCCL over xGMI) over  first-bucket size x bucket size x bucket dtype  and prints ONE table:
#   patches/s, ms/step, exposed (non-overlapped) all-reduce ms, backward-to-ready ms  per setting.
# The knobs are nndetection_amd/ddp.py's: NNDET_DDP_FIRST_MB (small first bucket so communication starts under the head backward),
# NNDET_DDP_BUCKET_MB, NNDET_DDP_BF16 (16-bit wire format: half the bytes per xGMI link, gradients rounded once).
#
#   tools/scale_sweep.sh [N=8] [steps=40] [warmup=10]        env: FIRST="1 4 8" BUCKET="12 24 48 76" BF16="0 1" PORT=29533
#   N=1 forces the RCCL / bucket path at world size 1 (NNDET_BENCH_FORCE_DIST=1): the bookkeeping cost without wire time.
#   DRY=1: rehearsal without GPUs -- the same launcher line with `bench.py --cpu-dry-run` (gloo, CPU stand-in network; prints no rate):
#   checks the rendezvous / per-rank seeds / bucket layout agreement / a positive-free rank end to end (tests/test_launch_rehearsal.py).
cd "$(dirname "$0")/.."
N=${1:-8}; STEPS=${2:-40}; WARM=${3:-10}
EXTRA="--no-extras"; [ "${DRY:-0}" = "1" ] && EXTRA="--cpu-dry-run"
FIRST=${FIRST:-"1 4 8"}; BUCKET=${BUCKET:-"12 24 48 76"}; BF16=${BF16:-"0 1"}; PORT=${PORT:-29533}
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
mkdir -p gpurun_out
OUT=gpurun_out/scale_sweep_n${N}.txt
printf "# bench.py --gpus %d --steps %d --warmup %d --no-extras; %s\n" "$N" "$STEPS" "$WARM" "$(date -u +%FT%TZ)" | tee "$OUT"
printf "%-9s %-10s %-5s %12s %10s %14s %16s\n" first_MB bucket_MB bf16 patches/s ms/step exposed_ar_ms bwd_to_ready_ms | tee -a "$OUT"
run() { # first bucket bf16
  local log=gpurun_out/.sweep_$$.json
  if [ "$N" -gt 1 ]; then
    NNDET_DDP_FIRST_MB=$1 NNDET_DDP_BUCKET_MB=$2 NNDET_DDP_BF16=$3 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" \
      --master-addr 127.0.0.1 --master-port "$PORT" bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARM" $EXTRA 2>/dev/null | grep '^{' | tail -1 > "$log"
  else
    NNDET_BENCH_FORCE_DIST=1 MASTER_PORT=$PORT NNDET_DDP_FIRST_MB=$1 NNDET_DDP_BUCKET_MB=$2 NNDET_DDP_BF16=$3 timeout 900 \
      python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARM" $EXTRA 2>/dev/null | grep '^{' | tail -1 > "$log"
  fi
  python - "$log" "$1" "$2" "$3" <<'PY' | tee -a "$OUT"
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    p = d.get("ddp") or {}
    if d.get("dry_run"):
        ok = d["params_identical_on_all_ranks"] and d["all_buckets_launched_from_hooks_on_all_ranks"] is not False
        print("%-9s %-10s %-5s %12s %10.3f %14s %16s" % (sys.argv[2], sys.argv[3], sys.argv[4], "dry-run:" + ("ok" if ok else "FAILED"), d["ms_per_step"],
              "%d buckets" % len(p.get("bucket_numel", [])), "identical" if d["params_identical_on_all_ranks"] else "DIVERGED"))
    else:
        print("%-9s %-10s %-5s %12.1f %10.3f %14s %16s" % (sys.argv[2], sys.argv[3], sys.argv[4], d["value"], d["ms_per_step"],
              p.get("exposed_allreduce_ms", "-"), p.get("backward_to_ready_ms", "-")))
except Exception as e:
    print("%-9s %-10s %-5s   failed: %s" % (sys.argv[2], sys.argv[3], sys.argv[4], e))
PY
  rm -f "$log"
}
for b16 in $BF16; do for f in $FIRST; do for b in $BUCKET; do run "$f" "$b" "$b16"; done; done; done
[ "${DRY:-0}" = "1" ] || echo "# baseline without the data-parallel path (N=1 only): $( [ "$N" -eq 1 ] && python bench.py --steps "$STEPS" --warmup "$WARM" --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' )" | tee -a "$OUT"
