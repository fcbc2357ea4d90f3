#!/bin/bash
# what the driver runs at round end, on one box: the GPU tests, smoke(), the default bench line
source tools/gpu_steps.sh
O=gpurun_out/r02
mkdir -p $O
step pytest 900 bash -c "python -m pytest tests -m gpu -q --maxfail=8 --durations=8 > $O/gputest_final.log 2>&1"
tail -14 $O/gputest_final.log
step smoke 300 bash -c "python -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1"
tail -2 $O/smoke.log
step bench 400 bash -c "python bench.py > $O/bench_final.json 2> $O/bench_final.err"
python - <<'PY'
import json
l=json.load(open("gpurun_out/r02/bench_final.json"))
r=l.pop("roofline")
print(json.dumps(l)[:1500])
print({k: r[k] for k in ("bound","achieved","peak","frac","traffic","kernel_ms")}, r["issue_model"]["ratio"])
PY
