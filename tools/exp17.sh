#!/bin/bash
out=gpurun_out/exp17; mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu > $out/gpu_all.txt 2>&1
