#!/bin/bash
# usage (GPU box): tools/experiments/quick_check.sh -- goldens through the four-wave kernel, full-size + soak-regression
# suites, then the shipped library's time at six batch sizes (tools/variants.py, 96 000 samples, config #3 signal)
out=gpurun_out/experiment; mkdir -p $out
FSKHIP_SPLIT=4 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden and four" > $out/parity.txt 2>&1
tail -2 $out/parity.txt
grep -q " passed" $out/parity.txt || exit 1
N=96000
{
for S in 4096 8192 32768 65536 81920 131072; do
echo "# $S x $N"
VAR_TIMEOUT=90 timeout 400 python tools/variants.py $S $N "auto" "auto2"
done
} > $out/variants.txt 2>&1
cat $out/variants.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_soak_regressions.py -x -q -m gpu > $out/tests_full.txt 2>&1
tail -2 $out/tests_full.txt
