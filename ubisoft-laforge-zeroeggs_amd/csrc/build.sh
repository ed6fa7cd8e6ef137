#!/bin/bash
# Build the MI355X engine library: every .hip translation unit -> zeggs/libzeggs_hip.so (gfx950 only).
set -e
cd "$(dirname "$0")"
# experiments: ZEGGS_DEFS="-DZEGGS_U=4" ZEGGS_OUT=../zeggs/libzeggs_alt.so bash build.sh ; load with ZEGGS_LIB=<path>
OUT=${ZEGGS_OUT:-../zeggs/libzeggs_hip.so}
BD=${ZEGGS_BUILD_DIR:-build}
mkdir -p $BD
pids=()
SRC="gemm gemm_split kernels attention encoders decoder decoder_fast decode_persistent train_persistent train_dual train_bwd_persistent loudness loss misc mel anim style_gru hostio funcs"
for f in $SRC; do
  if [ ! -f $BD/$f.o ] || [ $f.hip -nt $BD/$f.o ] || [ common.h -nt $BD/$f.o ] || [ decoder_ws.h -nt $BD/$f.o ] || [ dec_math.h -nt $BD/$f.o ] || [ ../../include/zeggs_hip.h -nt $BD/$f.o ] || [ kernels.h -nt $BD/$f.o ] || [ gemm.h -nt $BD/$f.o ] || [ tp_common.h -nt $BD/$f.o ] || [ dec_prologue.h -nt $BD/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $ZEGGS_DEFS -c $f.hip -o $BD/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
OBJ=""
for f in $SRC; do OBJ="$OBJ $BD/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJ
echo "built $OUT"
