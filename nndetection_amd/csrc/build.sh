#!/bin/bash
# Build libnndet_amd.so for gfx950 (cross-compiles without a GPU). Usage: csrc/build.sh [-j N]
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result"
mkdir -p _obj
pids=()
# build.sh --asm DIR : emit the gfx950 ISA of every source (same flags) into DIR/*.s instead of building the library
# (tests/test_isa_hazards.py scans it with tools/scan_store_hazard.py)
ASM_DIR=""
if [ "$1" = "--asm" ]; then ASM_DIR=$2; mkdir -p "$ASM_DIR"; fi
build() { # src extra-flags
  local src=$1; shift
  if [ -n "$ASM_DIR" ]; then
    $HIPCC $COMMON "$@" --cuda-device-only -S "$src" -o "$ASM_DIR/${src%.hip}.s" 2>/dev/null &
    pids+=($!)
    return
  fi
  local obj=_obj/${src%.hip}.o
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ common.h -nt "$obj" ] || [ conv_common.h -nt "$obj" ] || [ radix_select.h -nt "$obj" ] || [ ../../include/nndet_amd.h -nt "$obj" ]; then
    $HIPCC $COMMON "$@" -c "$src" -o "$obj" &
    pids+=($!)
  fi
}
# bit-exact box kernels: no FMA contraction (IEEE division / sqrt are hipcc defaults)
build nms3d.hip -ffp-contract=off
build boxes.hip -ffp-contract=off
build atss3d.hip -ffp-contract=off
build postproc.hip -ffp-contract=off
build targets.hip
build sampler.hip
build wbc3d.hip -ffp-contract=off
# k_ig3r is one fully unrolled 432-MFMA tile body: beyond the default size limit of '#pragma unroll'
build conv_igemm.hip -mllvm -pragma-unroll-threshold=1000000
build conv_wgrad.hip
build conv_pw.hip
build conv_dgs.hip
build conv_ig3s.hip
build conv_stem.hip
build norm.hip
build segloss.hip
build headio.hip
build sparse_out.hip
build segbranch.hip
build api.hip
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
[ $rc -eq 0 ] || { echo "compile failed"; exit 1; }
[ -z "$ASM_DIR" ] || { echo "ISA in $ASM_DIR"; exit 0; }
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o libnndet_amd.so _obj/*.o
echo "built $(pwd)/libnndet_amd.so"
