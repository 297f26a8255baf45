#!/bin/bash
# Diagnostic build of libaslhip.so with extra compiler flags, for same-box A/Bs:
#   bash tools/build_variant.sh <name> [-DFLAG=...]...   ->  build/<name>/libaslhip.so   (select it with ASL_LIB_PATH)
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/build/$NAME
mkdir -p $OUT
cd $ROOT/video_asl_recognition_amd/csrc
SRCS=$(sed -n 's/^SRCS = //p' Makefile)
for f in $SRCS; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wno-unused-function "$@" -c $f -o $OUT/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OUT/*.o -o $OUT/libaslhip.so
rm -f $OUT/*.o
echo built $OUT/libaslhip.so
