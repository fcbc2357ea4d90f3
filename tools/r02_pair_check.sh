#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r02
mkdir -p $O
step pytest_pair 900 bash -c "python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=5 -k 'align or overlap or matrices or pipeline or large_pair or c5 or cli or dropin or plain or beyond or degenerate' > $O/gputest_pair.log 2>&1"
tail -5 $O/gputest_pair.log
step g_dbg 120 bash -c "PWA_DEBUG=1 python bench.py --workload g --steps 3 --warmup 2 2>&1 | tail -16"
for w in g gb c2 c5; do
  step bench_$w 120 bash -c "python bench.py --workload $w --steps 10 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err"
  python - "$O/bench_$w.json" <<'PY'
import json,sys
l=json.load(open(sys.argv[1])); r=l["roofline"]
print(sys.argv[1].split("/")[-1], "value %.1f ms/step %.2f fill_ms %.3f tb_ms %.3f frac %.3f" % (l["value"], l["ms_per_step"], r["kernel_ms"], r.get("traceback_ms",0), r["frac"]), l.get("result"), l.get("verified_vs_cpu"), l.get("invalid"))
PY
done
