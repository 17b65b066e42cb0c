#!/bin/bash
# usage: tools/build_variant_all.sh <tag> [extra hipcc flags for EVERY translation unit ...]   (constants in shared headers)
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/webaudio_modem_amd/csrc
mkdir -p $root/tools/build/$tag
for f in fsk_api fsk_demod fsk_pipe fsk_blk fsk_mod fsk_xmodem fsk_processor fsk_fir; do
  extra=""; case $f in fsk_pipe|fsk_blk) extra="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $extra "$@" -c -o $root/tools/build/$tag/$f.o $csrc/$f.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/build/libfskhip_$tag.so $root/tools/build/$tag/*.o
echo built tools/build/libfskhip_$tag.so
