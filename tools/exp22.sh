#!/bin/bash
out=gpurun_out/exp22; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gpu_tests.txt 2>&1
tail -5 $out/gpu_tests.txt
