#!/bin/bash
out=gpurun_out/exp11; mkdir -p $out
N=96000
timeout 900 python tools/variants.py 65536 $N "blk@stamp:VAR_STAMPS=3,FSKHIP_SPLIT=4" > $out/variants.txt 2>&1
