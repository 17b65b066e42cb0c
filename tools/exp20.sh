#!/bin/bash
out=gpurun_out/exp20; mkdir -p $out
N=96000
{
for S in 4096 8192 65536; do
echo "# $S x $N"
timeout 900 python tools/variants.py $S $N "blk"
done
} > $out/variants.txt 2>&1
FSKHIP_SPLIT=4 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden and four" > $out/parity.txt 2>&1
C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"
timeout 600 bash tools/pmc.sh r03blk_insts "$C" --seconds 1 --steps 3 --warmup 1 --no-side > $out/pmc_insts_blk.txt 2>&1
rm -rf gpurun_out/pmc_r03*
