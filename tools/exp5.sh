#!/bin/bash
out=gpurun_out/exp5; mkdir -p $out
N=96000
{
for S in 8192 65536; do
echo "# blk3 vs pipe, $S x $N"
timeout 900 python tools/variants.py $S $N "pipe@stamp:VAR_STAMPS=1,FSKHIP_SPLIT=1" "blk3@stamp:VAR_STAMPS=1,FSKHIP_SPLIT=4" "blk3_nostamp:FSKHIP_SPLIT=4"
done
} > $out/variants.txt 2>&1
FSKHIP_SPLIT=4 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $out/parity_blk3.txt 2>&1
