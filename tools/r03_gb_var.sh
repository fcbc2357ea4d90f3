#!/bin/bash
# how much does the gb fill vary on one box: three plain processes, then one under rocprofv3 --kernel-trace --stats, then one plain again
source tools/gpu_steps.sh
export TMPDIR=/tmp
O=gpurun_out/r03
mkdir -p $O
line() { python3 -c "
import json
l = json.loads([x for x in open('$O/probe.json').read().splitlines() if x.startswith('{')][-1]); r = l['roofline']
print('$1: fill %.3f ms  frac %.3f  kernel GCUPS %.0f' % (r['kernel_ms'], r['frac'], r['kernel_gcups']))"; }
for i in 1 2 3; do step gb$i 200 python3 bench.py --workload gb --steps 10 --warmup 2 > $O/probe.json 2>/dev/null; line "plain run $i"; done
rm -rf $O/prof_gbv; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_gbv -- python3 bench.py --workload gb --steps 10 --warmup 2 > $O/probe.json 2>/dev/null; line "under rocprofv3"
f=$(find $O/prof_gbv -name "*kernel_stats.csv" | head -1); grep mini_fill $f | cut -c1-200
step gb4 200 python3 bench.py --workload gb --steps 10 --warmup 2 > $O/probe.json 2>/dev/null; line "plain run 4"
