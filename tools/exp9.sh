#!/bin/bash
out=gpurun_out/exp9; mkdir -p $out
N=96000
{
for S in 49152 65536; do
echo "# blk (4 waves), $S x $N"
timeout 900 python tools/variants.py $S $N "blk@stamp:VAR_STAMPS=1,FSKHIP_SPLIT=4"
done
} > $out/variants.txt 2>&1
