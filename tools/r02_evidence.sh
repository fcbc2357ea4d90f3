#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r02
mkdir -p $O
step scal_rl4 200 bash -c "PWA_FORCE_RL=4 python tools/pair_scaling.py nw > $O/scaling_rl4.txt 2>&1"
step scal_rl2 200 bash -c "PWA_FORCE_RL=2 python tools/pair_scaling.py nw > $O/scaling_rl2.txt 2>&1"
step scal_sw 200 bash -c "PWA_FORCE_RL=2 python tools/pair_scaling.py sw > $O/scaling_sw_rl2.txt 2>&1"
step stamps0 200 bash -c "PWA_TRACE_STRIPE=0 PWA_STAMPS=$O/stamps_c5_a.txt python bench.py --workload c5 --steps 1 --warmup 1 > /dev/null 2>&1"
step stamps90 200 bash -c "PWA_TRACE_STRIPE=200 PWA_STAMPS=$O/stamps_c5_b.txt python bench.py --workload c5 --steps 1 --warmup 1 > /dev/null 2>&1"
(echo "== C5 (NW 100k x 100k, RL = 4, W = 4), stripes 0-3 traced"; python tools/stamps_summary.py $O/stamps_c5_a.txt; echo; echo "== same, stripes 200-203 traced"; python tools/stamps_summary.py $O/stamps_c5_b.txt) > $O/stripe_stamps.txt 2>&1
step cold 300 bash -c "python tools/cold_start.py > $O/cold_start.txt 2>&1"
cat $O/stripe_stamps.txt | cut -c1-250
