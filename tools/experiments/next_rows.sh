#!/bin/bash
out=gpurun_out/experiment; mkdir -p $out
timeout 600 python tools/bench_next.py > $out/next_rows.jsonl 2>&1
cat $out/next_rows.jsonl | cut -c1-260
