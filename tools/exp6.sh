#!/bin/bash
out=gpurun_out/exp6; mkdir -p $out
N=96000
{
for S in 8192 65536; do
echo "# blk (4 waves) vs pipe, $S x $N"
timeout 900 python tools/variants.py $S $N "pipe@stamp:VAR_STAMPS=1,FSKHIP_SPLIT=1" "blk@stamp:VAR_STAMPS=1,FSKHIP_SPLIT=4" "blk_nostamp:FSKHIP_SPLIT=4"
done
} > $out/variants.txt 2>&1
FSKHIP_SPLIT=4 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak_regressions.py -x -q -m gpu > $out/parity_blk.txt 2>&1
timeout 1800 python -m pytest tests -x -q -m gpu > $out/gpu_all.txt 2>&1
