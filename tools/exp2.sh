#!/bin/bash
# round-3 experiment 2: what the access pattern allows (read_probe), calibration of this box against the r02 headline,
# ablations with the hook fixed
out=gpurun_out/exp2; mkdir -p $out
timeout 600 ./tools/build/read_probe 65536 49152 > $out/read_probe.txt 2>&1
N=96000
{
echo "# ablation, two-wave kernel, 65536 x $N (bit w = wave w skips its arithmetic)"
timeout 900 python tools/variants.py 65536 $N "pipe" "abl0@abl:FSK_ABLATE=0" "abl_front_only@abl:FSK_ABLATE=2" "abl_back_only@abl:FSK_ABLATE=1" "abl_skeleton@abl:FSK_ABLATE=3"
echo "# ablation, three-wave kernel (3 slots), 65536 x $N"
timeout 1500 python tools/variants.py 65536 $N "abl0@abl_s3:FSKHIP_SPLIT=3,FSK_ABLATE=0" "w0_only@abl_s3:FSKHIP_SPLIT=3,FSK_ABLATE=6" "w1_only@abl_s3:FSKHIP_SPLIT=3,FSK_ABLATE=5" "w2_only@abl_s3:FSKHIP_SPLIT=3,FSK_ABLATE=3" "skeleton@abl_s3:FSKHIP_SPLIT=3,FSK_ABLATE=7" "w01@abl_s3:FSKHIP_SPLIT=3,FSK_ABLATE=4"
} > $out/variants.txt 2>&1
timeout 900 python bench.py --seconds 2 --steps 5 --warmup 2 --no-side --cpu-seconds 3 > $out/bench_2s.txt 2>&1
