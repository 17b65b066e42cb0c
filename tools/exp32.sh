#!/bin/bash
out=gpurun_out/exp32; mkdir -p $out
N=96000
{
for S in 66560 69632 73728 81920 90112 98304 114688 131072 147456 262144 524288; do
echo "# $S x $N"
VAR_TIMEOUT=90 timeout 600 python tools/variants.py $S $N "auto" "s768:FSKHIP_SLICE_TILES=768"
done
} > $out/variants.txt 2>&1
cat $out/variants.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "time_sliced or chunking" > $out/tests.txt 2>&1
tail -2 $out/tests.txt
