#!/bin/bash
out=gpurun_out/exp18; mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu -x > $out/gpu_all.txt 2>&1
timeout 900 python bench.py --workload c5 --streams 16384 --steps 5 --warmup 1 --cpu-seconds 6 > $out/bench_c5.txt 2>&1
timeout 900 python bench.py --workload c2 --streams 4096 --no-side --steps 5 > $out/bench_c2.txt 2>&1
timeout 900 python bench.py --workload c4 --streams 32768 --steps 4 --cpu-seconds 6 > $out/bench_c4.txt 2>&1
timeout 900 python bench.py --precision f64 --seconds 10 --steps 2 --warmup 1 --no-side --cpu-seconds 0 > $out/bench_f64_full.txt 2>&1
