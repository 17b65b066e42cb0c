#!/bin/bash
out=gpurun_out/exp24; mkdir -p $out
N=96000
{
for S in 24576 40960 49152 57344 65536 69632 81920; do
echo "# $S x $N"
VAR_TIMEOUT=60 timeout 400 python tools/variants.py $S $N "y6:FSKHIP_BLK_YSLOTS=6" "auto"
done
} > $out/variants.txt 2>&1
cat $out/variants.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1
tail -5 $out/gpu_tests.txt
