#!/bin/bash
# Build libwseg_hip.so in-tree for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libwseg_hip.so
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17"
SRCS=$(ls *.hip)
mkdir -p _obj
pids=()
for f in $SRCS; do
  o=_obj/${f%.hip}.o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ] || [ ../../include/wseg_hip.h -nt "$o" ]; then
    hipcc $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 --hip-link -shared -fPIC _obj/*.o -o $OUT
echo "built $OUT"
