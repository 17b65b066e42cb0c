#!/bin/bash
# time-sliced persistent launch: parity first, then the sweep across the cliff
out=gpurun_out/experiment; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "time_sliced" > $out/tests_small.txt 2>&1
tail -3 $out/tests_small.txt
grep -q passed $out/tests_small.txt || exit 1
N=96000
{
for S in 65536 81920 98304 114688 131072 196608; do
echo "# $S x $N"
VAR_TIMEOUT=60 timeout 300 python tools/variants.py $S $N "off:FSKHIP_SLICE_TILES=off" "s768" "s192:FSKHIP_SLICE_TILES=192" "s384:FSKHIP_SLICE_TILES=384"
done
} > $out/variants.txt 2>&1
cat $out/variants.txt
