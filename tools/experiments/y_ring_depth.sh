#!/bin/bash
# y-ring depth sweep
out=gpurun_out/experiment; mkdir -p $out
N=96000
{
for S in 4096 8192 16384 32768 49152; do
echo "# $S x $N"
VAR_TIMEOUT=60 timeout 400 python tools/variants.py $S $N "y6:FSKHIP_BLK_YSLOTS=6" "y8:FSKHIP_BLK_YSLOTS=8" "y12:FSKHIP_BLK_YSLOTS=12" "y16:FSKHIP_BLK_YSLOTS=16" "y24:FSKHIP_BLK_YSLOTS=24" "auto"
done
for S in 65536 131072; do
echo "# $S x $N"
VAR_TIMEOUT=60 timeout 400 python tools/variants.py $S $N "y6:FSKHIP_BLK_YSLOTS=6" "y7:FSKHIP_BLK_YSLOTS=7" "y8:FSKHIP_BLK_YSLOTS=8" "auto"
done
} > $out/variants.txt 2>&1
cat $out/variants.txt
FSKHIP_SPLIT=4 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden and four" > $out/parity.txt 2>&1
tail -2 $out/parity.txt
timeout 300 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "time_sliced or ragged" > $out/tests_small.txt 2>&1
tail -2 $out/tests_small.txt
