#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r02
mkdir -p $O
step pytest 900 bash -c "python -m pytest tests -m gpu -q --maxfail=8 --durations=15 > $O/gputest.log 2>&1"
tail -25 $O/gputest.log
step bench_c3 300 bash -c "python bench.py > $O/bench_c3.json 2> $O/bench_c3.err"
step valu_issue 200 bash -c "tools/valu_issue > $O/valu_issue.txt 2>&1"
step bench_dist1 200 bash -c "BENCH_FORCE_DIST=1 python bench.py --steps 3 --no-cpu-baseline > $O/bench_c3_dist1.json 2> $O/bench_c3_dist1.err"
