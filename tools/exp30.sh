#!/bin/bash
out=gpurun_out/exp30; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1
tail -3 $out/gpu_tests.txt
timeout 900 python tools/soak.py 420 7301 > $out/soak_a.txt 2>&1
tail -4 $out/soak_a.txt
