#!/bin/bash
# Build libzeggs_hip.so for gfx950 (cross-compiles without a GPU). Output lands in ../zeggs/ (in-tree).
set -e
cd "$(dirname "$0")"
OUT=../zeggs/libzeggs_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result ${ZEGGS_HIPCC_FLAGS}"
mkdir -p build
pids=()
for f in gemm kernels encoders decoder decoder_fast loss misc mel; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ gemm.h -nt build/$f.o ] \
     || [ kernels.h -nt build/$f.o ] || [ decoder_ws.h -nt build/$f.o ] || [ dec_math.h -nt build/$f.o ] || [ ../../include/zeggs_hip.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o $OUT
echo "built $OUT"
