#!/bin/bash
# dev tool (runs here, cross-compiles): variants of libcagroup3d_hip.so whose tile kernel is built with -DT2_DBG=<n> (knock-outs,
# see spconv_tile2.hip) -> cagroup3d_amd/csrc/dev/libcg3d_dbg<n>.so (git-ignored, travels with gpurun).
# usage: bash tools/build_tile_dbg.sh 1 2 3 4 8 16 ...     [SRC=spconv_tile2.hip] [DEF=T2_DBG]
cd "$(dirname "$0")/../cagroup3d_amd/csrc" || exit 1
python build.py > /dev/null || exit 1
SRC=${SRC:-spconv_tile2.hip}; DEF=${DEF:-T2_DBG}; OBJ=${SRC%.hip}.o
mkdir -p dev
for n in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics -D$DEF=$n -c $SRC -o dev/${OBJ%.o}_$n.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o dev/libcg3d_dbg$n.so $(ls *.o | grep -v "^$OBJ\$") dev/${OBJ%.o}_$n.o && echo built dev/libcg3d_dbg$n.so ) &
done
wait
