#!/bin/bash
# usage: tools/build_lag.sh <lag> [extra flags] -- tools/build/libfskhip_z<lag>.so = the WHOLE library with FSK_ZLAG = <lag>
# (kZeroLagPairs, fsk_params.h: every translation unit sees it).  Measurement builds for tools/variants.py.
set -e
lag=$1; shift; tag=z$lag; case "$1" in h*) tag=z${lag}$1; set -- -DFSK_HLAG=${1#h} "${@:2}";; esac
tag=${TAG:-$tag}
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/webaudio_modem_amd/csrc
d=$root/tools/build/$tag
mkdir -p $d
for f in $csrc/fsk_*.hip; do
  b=$(basename $f .hip)
  slp=""; case $b in fsk_pipe|fsk_blk|fsk_blk6) slp="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $slp -DFSK_ZLAG=$lag "$@" -c -o $d/$b.o $f &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/build/libfskhip_$tag.so $d/*.o
echo built tools/build/libfskhip_$tag.so
