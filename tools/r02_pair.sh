#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r02
mkdir -p $O
step pytest_pair 900 bash -c "python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=5 -k 'align or overlap or matrices or pipeline or large_pair or c5 or cli or dropin or plain or beyond or degenerate' > $O/gputest_pair.log 2>&1"
tail -8 $O/gputest_pair.log
for w in c2 c5; do
  step bench_$w 120 bash -c "python bench.py --workload $w --steps 10 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err"
  step bench_${w}_rl2 120 bash -c "PWA_FORCE_RL=2 python bench.py --workload $w --steps 10 --warmup 2 > $O/bench_${w}_rl2.json 2> $O/bench_${w}_rl2.err"
  step bench_${w}_rl4 120 bash -c "PWA_FORCE_RL=4 python bench.py --workload $w --steps 10 --warmup 2 > $O/bench_${w}_rl4.json 2> $O/bench_${w}_rl4.err"
done
step bench_g 120 bash -c "python bench.py --workload g --steps 5 --warmup 2 > $O/bench_g.json 2> $O/bench_g.err"
step bench_gb 120 bash -c "python bench.py --workload gb --steps 5 --warmup 2 > $O/bench_gb.json 2> $O/bench_gb.err"
step valu_issue 200 bash -c "tools/valu_issue > $O/valu_issue.txt 2>&1"
step bench_dist1 200 bash -c "BENCH_FORCE_DIST=1 python bench.py --steps 3 --no-cpu-baseline > $O/bench_c3_dist1.json 2> $O/bench_c3_dist1.err"
for f in $O/bench_c2*.json $O/bench_c5*.json $O/bench_g.json $O/bench_gb.json; do python - "$f" <<'PY'
import json,sys
try:
    l=json.load(open(sys.argv[1])); r=l["roofline"]
    print(sys.argv[1].split("/")[-1], "value %.1f ms/step %.2f fill_ms %.3f tb_ms %.3f frac %.3f" % (l["value"], l["ms_per_step"], r["kernel_ms"], r.get("traceback_ms",0), r["frac"]), l.get("result"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
