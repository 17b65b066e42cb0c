#!/bin/bash
# call length at which the block kernel overtakes round 2's kernels (launch after launch of N samples, state carried on)
out=gpurun_out/experiment; mkdir -p $out
{
for S in 262144 65536 8192; do
for N in 128 256 512 1024 2048 4096; do
echo "# $S x $N"
VAR_STEPS=20 VAR_TIMEOUT=90 timeout 300 python tools/variants.py $S $N "blk:FSKHIP_SPLIT=4" "r02:FSKHIP_SPLIT=b"
done
done
} > $out/short_calls.txt 2>&1
cat $out/short_calls.txt
