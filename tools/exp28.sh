#!/bin/bash
out=gpurun_out/exp28; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1
tail -3 $out/gpu_tests.txt
N=96000
{
for S in 4096 8192 16384 32768 40960 49152 65536 81920 98304 131072 262144; do
echo "# $S x $N"
VAR_TIMEOUT=90 timeout 400 python tools/variants.py $S $N "auto"
done
} > $out/variants.txt 2>&1
cat $out/variants.txt
