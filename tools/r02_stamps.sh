#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r02
mkdir -p $O
step stamps 200 bash -c "PWA_STAMPS=$O/stamps_c5.txt python bench.py --workload c5 --steps 1 --warmup 1 > $O/bench_c5_stamps.json 2> $O/bench_c5_stamps.err"
head -40 $O/stamps_c5.txt
