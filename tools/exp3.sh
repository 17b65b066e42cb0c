#!/bin/bash
out=gpurun_out/exp3; mkdir -p $out
N=96000
{
echo "# wave stamps, 65536 x $N"
timeout 900 python tools/variants.py 65536 $N "pipe" "pipe_stamp@stamp:VAR_STAMPS=1" "pipe3_s3_stamp@stamp_s3:VAR_STAMPS=1,FSKHIP_SPLIT=3" "pipe3_s3@s3:FSKHIP_SPLIT=3"
echo "# wave stamps, 8192 x $N"
timeout 900 python tools/variants.py 8192 $N "pipe_stamp@stamp:VAR_STAMPS=1,FSKHIP_SPLIT=1" "pipe3_s3_stamp@stamp_s3:VAR_STAMPS=1,FSKHIP_SPLIT=3"
echo "# wave stamps, 32768 x $N"
timeout 900 python tools/variants.py 32768 $N "pipe_stamp@stamp:VAR_STAMPS=1,FSKHIP_SPLIT=1" "pipe3_s3_stamp@stamp_s3:VAR_STAMPS=1,FSKHIP_SPLIT=3"
} > $out/variants.txt 2>&1
