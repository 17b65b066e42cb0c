#!/bin/bash
# which waves get the SIMD first
out=gpurun_out/experiment; mkdir -p $out
N=96000
{
for S in 32768 65536 131072; do
echo "# $S x $N"
VAR_TIMEOUT=90 timeout 600 python tools/variants.py $S $N "base" "prio2@prio2" "prio3@prio3" "per128@per128" "per256@per256" "base2"
done
} > $out/variants.txt 2>&1
cat $out/variants.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "write_back or y_ring" > $out/tests.txt 2>&1
tail -3 $out/tests.txt
