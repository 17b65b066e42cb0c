#!/bin/bash
# usage (GPU box): tools/batch_sweep.sh > gpurun_out/batch_sweep.txt
# Throughput of the library's own kernel choice over the batch size (BASELINE config #3's signal, 2 s per stream).
for S in 4096 8192 16384 32768 40960 49152 65536 69632 81920 98304 114688 131072 196608 262144 524288; do
  secs=2; if [ $S -ge 524288 ]; then secs=1; fi
  line=$(timeout 300 python bench.py --streams $S --seconds $secs --steps 3 --warmup 1 --no-side --cpu-seconds 0 2>/dev/null | tail -n 1)
  echo "$line" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%7d streams  %9.1f Msamples/s  %5.1f %% of 8 TB/s  %s' % (d['config']['streams_per_gpu'], d['value'], 100 * d['roofline']['frac'], d['roofline']['kernel']))"
done
# the small batches again at BASELINE config #3's own length (10 s per stream: launch start-up and drain amortised)
for S in 4096 8192 16384; do
  line=$(timeout 300 python bench.py --streams $S --seconds 10 --steps 3 --warmup 1 --no-side --cpu-seconds 0 2>/dev/null | tail -n 1)
  echo "$line" | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%7d streams  %9.1f Msamples/s  %5.1f %% of 8 TB/s  %s  (10 s per stream)' % (d['config']['streams_per_gpu'], d['value'], 100 * d['roofline']['frac'], d['roofline']['kernel']))"
done
