#!/bin/bash
# what the driver runs at round end, on one box: the GPU tests, smoke(), the default bench line (plain and under torch.distributed.run)
source tools/gpu_steps.sh
O=gpurun_out/r03
mkdir -p $O
step pytest 900 bash -c "python3 -m pytest tests -m gpu -q --maxfail=8 --durations=8 > $O/gputest_final.log 2>&1"
tail -14 $O/gputest_final.log
step smoke 300 bash -c "python3 -c 'import __graft_entry__ as g; g.smoke()' > $O/smoke.log 2>&1"
tail -2 $O/smoke.log
step bench 400 bash -c "python3 bench.py > $O/bench_final.json 2> $O/bench_final.err"
step bench_trun 400 bash -c "python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 1 > $O/bench_final_trun.json 2> $O/bench_final_trun.err"
python3 - <<'PY'
import json
for f in ("bench_final.json", "bench_final_trun.json"):
    l=json.loads([x for x in open("gpurun_out/r03/" + f).read().splitlines() if x.startswith("{")][-1])
    r=l.pop("roofline")
    print(f, json.dumps({k: l[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","scaling","dtype")}), l.get("invalid",""), l.get("checksum"), l.get("verified_vs_cpu"), (l.get("cpu_baseline") or {}).get("value"), l.get("dist"))
    print({k: r[k] for k in ("bound","achieved","peak","frac","traffic","kernel_ms")}, r["issue_model"]["ratio"])
PY
