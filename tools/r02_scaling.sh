#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r02
mkdir -p $O
step scal_rl4 200 bash -c "PWA_FORCE_RL=4 python tools/pair_scaling.py nw > $O/scaling_rl4.txt 2>&1"
step scal_rl2 200 bash -c "PWA_FORCE_RL=2 python tools/pair_scaling.py nw > $O/scaling_rl2.txt 2>&1"
step scal_rl4_w1 200 bash -c "PWA_FORCE_RL=4 PWA_FORCE_W=1 python tools/pair_scaling.py nw > $O/scaling_rl4_w1.txt 2>&1"
cat $O/scaling_rl4.txt $O/scaling_rl2.txt $O/scaling_rl4_w1.txt
