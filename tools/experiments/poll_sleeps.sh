#!/bin/bash
# hand-off poll sleeps
out=gpurun_out/experiment; mkdir -p $out
N=96000
{
for S in 8192 65536 131072; do
echo "# $S x $N"
VAR_TIMEOUT=90 timeout 600 python tools/variants.py $S $N "base" "a4@a4" "a8@a8" "a4bc2@a4bc2" "a8bc4@a8bc4" "a0@a0" "base2"
done
} > $out/variants.txt 2>&1
cat $out/variants.txt
