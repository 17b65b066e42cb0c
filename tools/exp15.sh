#!/bin/bash
out=gpurun_out/exp15; mkdir -p $out
{
for S in 4096 16384 49152 65536 81920 98304 131072 196608 262144; do
N=96000; if [ $S -gt 131072 ]; then N=48000; fi
echo "# $S x $N"
timeout 900 python tools/variants.py $S $N "default" "blk:FSKHIP_SPLIT=4"
done
} > $out/sweep.txt 2>&1
