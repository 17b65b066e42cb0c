#!/bin/bash
out=gpurun_out/experiment; mkdir -p $out
N=96000
{
for S in 8192 65536; do
echo "# $S x $N"
VAR_TIMEOUT=90 VAR_STAMPS=1 timeout 400 python tools/variants.py $S $N "stamp@stamp" "plain" "plain2"
done
} > $out/variants.txt 2>&1
cat $out/variants.txt
