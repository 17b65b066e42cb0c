#!/bin/bash
out=gpurun_out/exp14; mkdir -p $out
N=96000
{
echo "# ring depth experiment, 8192 x $N (LDS per group grows: timing experiment only)"
timeout 900 python tools/variants.py 8192 $N "blk_s6@stamp:VAR_STAMPS=1,FSKHIP_SPLIT=4" "blk_s8@s8:VAR_STAMPS=1,FSKHIP_SPLIT=4" "blk_s10@s10:VAR_STAMPS=1,FSKHIP_SPLIT=4"
} > $out/variants.txt 2>&1
