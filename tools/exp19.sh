#!/bin/bash
out=gpurun_out/exp19; mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu > $out/gpu_all.txt 2>&1
timeout 900 python bench.py --workload c5 --streams 16384 --steps 5 --warmup 1 --cpu-seconds 6 > $out/bench_c5.txt 2>&1
