#!/bin/bash
out=gpurun_out/exp36; mkdir -p $out
timeout 1000 python tools/soak.py 780 4242 > $out/soak_c.txt 2>&1
tail -3 $out/soak_c.txt
