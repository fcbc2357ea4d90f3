#!/bin/bash
# mid-sized patterns in batches: one pair per wave (mini-stripe kernels, LN = 64) against the stripe engine's pipelined stripes (PWA_TB_ENGINE=0)
source tools/gpu_steps.sh
O=gpurun_out/r03
mkdir -p $O
line() { python3 -c "
import json
l = json.loads([x for x in open('$O/probe.json').read().splitlines() if x.startswith('{')][-1]); r = l['roofline']
print('$1: fill %.3f ms walk %.3f ms step %.3f ms kernel GCUPS %.0f frac %.3f written/alg %.2f %s %s' % (r['kernel_ms'], r['traceback_ms'], l['ms_per_step'], r['kernel_gcups'], r['frac'], r['written_over_algorithmic'], r['kernel'], l.get('invalid','')))"; }
for plen in 300 500 1000; do
  for eng in wide stripes; do
    if [ $eng = stripes ]; then export PWA_TB_ENGINE=0; else unset PWA_TB_ENGINE; fi
    step g 200 python3 bench.py --workload g --plen $plen --pairs 1024 --steps 5 --warmup 2 > $O/probe.json 2>/dev/null; line "g  1024 pairs $plen x 10k $eng"
    step gb 200 python3 bench.py --workload gb --plen $plen --pairs 1024 --steps 3 --warmup 1 > $O/probe.json 2>/dev/null; line "gb 1024 pairs $plen x 10k $eng"
  done
done
unset PWA_TB_ENGINE
