#!/bin/bash
# usage: tools/build_variant.sh <tag> [extra hipcc flags for fsk_pipe.hip and fsk_blk.hip ...]
# Builds tools/build/libfskhip_<tag>.so = the library with the whole-tile kernels compiled with extra flags (measurement
# builds; loaded by tools/variants.py through _lib.LIB_PATH, never by the package itself).
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/webaudio_modem_amd/csrc
make -C $csrc > /dev/null
mkdir -p $root/tools/build
for f in fsk_pipe fsk_blk fsk_blk6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -fno-slp-vectorize "$@" \
    -c -o $root/tools/build/${f}_$tag.o $csrc/$f.hip &
done
wait
objs=$(ls $csrc/build/fsk_*.o | grep -v "fsk_pipe\|fsk_blk")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/build/libfskhip_$tag.so $objs $root/tools/build/fsk_pipe_$tag.o $root/tools/build/fsk_blk_$tag.o $root/tools/build/fsk_blk6_$tag.o
echo built tools/build/libfskhip_$tag.so
