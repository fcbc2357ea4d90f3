#!/bin/bash
# A/B of the gb fill on ONE box, alternating processes: the in-tree library against build/libpwalign_oldmini.so = the same objects with
# mini_kernels.o compiled from the commit before the packed first maxima (git archive <commit>~1 bioinformatics-algorithms_amd/csrc |
# tar -x -C /tmp/old; hipcc ... -c /tmp/old/.../mini_kernels.hip; link with the current objects).  Result: profiles/r03_gb_fill_box_variance.txt
source tools/gpu_steps.sh
O=gpurun_out/r03
mkdir -p $O
line() { python3 -c "
import json
l = json.loads([x for x in open('$O/probe.json').read().splitlines() if x.startswith('{')][-1]); r = l['roofline']
print('$1: fill %.3f ms  frac %.3f %s' % (r['kernel_ms'], r['frac'], l.get('invalid','')))"; }
for rep in 1 2 3 4; do
  for v in new old; do
    if [ $v = old ]; then export PWA_LIB=$PWD/build/libpwalign_oldmini.so; else unset PWA_LIB; fi
    step gb 200 python3 bench.py --workload gb --steps 8 --warmup 2 --no-cpu-baseline > $O/probe.json 2>/dev/null; line "gb $v $rep"
  done
done
