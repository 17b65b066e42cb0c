#!/bin/bash
# the FSKProcessor quantum loop (128-sample launches at 262 144 streams): the library's own choice, round 2's kernels
# pinned, and -- when tools/build/libfskhip_r02.so exists (round 2's sources built apart) -- round 2's library
out=gpurun_out/experiment; mkdir -p $out
run() {  # $1 = FSKHIP_SPLIT, $2 = library or -
  FSKHIP_SPLIT=$1 timeout 300 python -c "
import sys, runpy
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
import envopts; envopts.install()     # FSKHIP_* -> fskhip_set_option
import webaudio_modem_amd._lib as L
if '$2' != '-': L.LIB_PATH = '$2'
sys.argv = ['bench_next.py', '--quanta', '100']
runpy.run_path('tools/bench_next.py', run_name='__main__')" 2>&1 | grep "FSKProcessor" | head -4
}
{
for sp in a 1 0; do echo "# FSKHIP_SPLIT=$sp"; run $sp -; done
if [ -f tools/build/libfskhip_r02.so ]; then echo "# round-2 library, its own choice"; run a tools/build/libfskhip_r02.so; fi
} > $out/processor.txt 2>&1
cat $out/processor.txt
