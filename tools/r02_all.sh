#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r02
mkdir -p $O
step pytest 900 bash -c "python -m pytest tests -m gpu -q --maxfail=8 > $O/gputest.log 2>&1"
tail -4 $O/gputest.log
step cli 900 bash -c "bash tools/r02_cli.sh > $O/cli_scale.txt 2>&1"
grep -E "real|arena|sum over|validate|sort|slot|uploads|engine" $O/cli_scale.txt | head -40
for w in c3 g; do
  step bench_$w 300 bash -c "python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err"
done
python - <<'PY'
import json
l=json.load(open("gpurun_out/r02/bench_c3.json")); print("c3", l["value"], l["ms_per_step"], l["host_call_inclusive"])
l=json.load(open("gpurun_out/r02/bench_g.json")); print("g", l["value"], l["ms_per_step"], l["roofline"]["kernel_ms"])
PY
