#!/bin/bash
out=gpurun_out/exp16; mkdir -p $out
timeout 1800 python -m pytest tests -x -q -m gpu > $out/gpu_all.txt 2>&1
timeout 400 python tools/soak.py 240 4242 > $out/soak_a.txt 2>&1
timeout 400 python tools/soak.py 240 777 > $out/soak_b.txt 2>&1
timeout 900 python bench.py > $out/bench.txt 2>&1
