#!/bin/bash
# own-span tiles on the block path: parity first, then timing
out=gpurun_out/experiment; mkdir -p $out
FSKHIP_SPLIT=4 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden and four" > $out/parity.txt 2>&1
tail -2 $out/parity.txt
grep -q " passed" $out/parity.txt || exit 1
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_soak_regressions.py -x -q -m gpu > $out/tests_full.txt 2>&1
tail -2 $out/tests_full.txt
N=96000
{
for S in 4096 8192 32768 65536 81920; do
echo "# $S x $N"
VAR_TIMEOUT=90 VAR_STAMPS=1 timeout 400 python tools/variants.py $S $N "plain" "stamp@stamp"
done
} > $out/variants.txt 2>&1
cat $out/variants.txt
