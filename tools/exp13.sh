#!/bin/bash
out=gpurun_out/exp13; mkdir -p $out
N=96000
{
for S in 8192 65536; do
echo "# priority rotation, $S x $N"
timeout 900 python tools/variants.py $S $N "blk_noprio@noprio:FSKHIP_SPLIT=4" "blk_prio:FSKHIP_SPLIT=4" "blk_prio_stamp@stamp:VAR_STAMPS=3,FSKHIP_SPLIT=4"
done
} > $out/variants.txt 2>&1
