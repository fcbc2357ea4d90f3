#!/bin/bash
# rocprofv3 summaries + PMC traffic of HEAD for the given workloads (tools/profile.sh), tag r03
source tools/gpu_steps.sh
for w in "$@"; do
  step prof_$w 500 bash tools/profile.sh $w r03
done
ls gpurun_out/profiles/
