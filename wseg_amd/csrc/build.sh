#!/bin/bash
# Build libwseg_hip.so in-tree for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=${WSEG_OUT:-../libwseg_hip.so}          # (development: WSEG_OUT=../libwseg_hip_probe.so WSEG_OBJ=_obj_probe keeps a probe build beside the product)
OBJ=${WSEG_OBJ:-_obj}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17"
# WSEG_PROBES=1: also compile the development probes (gemm256_probe, the 256x128 conv tile, the WSEG_WGRAD_DIAG / zero-page timing
# diagnostics, which give wrong results by design) — never part of the product library
if [ "${WSEG_PROBES:-0}" = "1" ]; then FLAGS="$FLAGS -DWSEG_PROBES"; fi
# WSEG_PROBES=2: the probes plus per-slot cycle sums inside the conv main loop (serialises what the real kernel overlaps: shares only)
if [ "${WSEG_PROBES:-0}" = "2" ]; then FLAGS="$FLAGS -DWSEG_PROBES -DWSEG_SLOTS"; fi
FLAGS="$FLAGS ${WSEG_EXTRA_FLAGS:-}"
SRCS=$(ls *.hip)
mkdir -p $OBJ
pids=()
for f in $SRCS; do
  o=$OBJ/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ conv_wgrad_kernels.h -nt "$o" ] || [ "$(cat $OBJ/.flags 2>/dev/null)" != "$FLAGS" ] || [ common.h -nt "$o" ] || [ ../../include/wseg_hip.h -nt "$o" ]; then
    hipcc $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
echo "$FLAGS" > $OBJ/.flags
hipcc --offload-arch=gfx950 --hip-link -shared -fPIC $OBJ/*.o -o $OUT
echo "built $OUT"
