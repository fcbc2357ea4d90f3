#!/bin/bash
source tools/gpu_steps.sh
for w in "$@"; do
  step prof_$w 500 bash tools/profile.sh $w r02
done
ls gpurun_out/profiles/
