#!/bin/bash
# every bench workload once, lines collected in gpurun_out/r03/bench_lines.jsonl (copied to profiles/r03_bench_lines.jsonl)
source tools/gpu_steps.sh
O=gpurun_out/r03
mkdir -p $O
: > $O/bench_lines.jsonl
: > $O/bench_lines.err
step c3 400 bash -c "python3 bench.py >> $O/bench_lines.jsonl 2>> $O/bench_lines.err"
for w in c3i long c4 hw3 hw4; do
  step $w 400 bash -c "python3 bench.py --workload $w >> $O/bench_lines.jsonl 2>> $O/bench_lines.err"
done
step c3i_nw 400 bash -c "python3 bench.py --workload c3i --nw >> $O/bench_lines.jsonl 2>> $O/bench_lines.err"
step long1 200 bash -c "python3 bench.py --workload long --pairs 1 --steps 20 --warmup 3 >> $O/bench_lines.jsonl 2>> $O/bench_lines.err"
for w in c2 c2b c5 g gb; do
  step $w 200 bash -c "python3 bench.py --workload $w --steps 10 --warmup 2 >> $O/bench_lines.jsonl 2>> $O/bench_lines.err"
done
step g8192 200 bash -c "python3 bench.py --workload g --pairs 8192 --steps 10 --warmup 2 >> $O/bench_lines.jsonl 2>> $O/bench_lines.err"
step gb128 200 bash -c "python3 bench.py --workload gb --plen 128 --steps 10 --warmup 2 >> $O/bench_lines.jsonl 2>> $O/bench_lines.err"
step dist1 200 bash -c "BENCH_FORCE_DIST=1 python3 bench.py --steps 3 --no-cpu-baseline >> $O/bench_lines.jsonl 2>> $O/bench_lines.err"
step distg 200 bash -c "BENCH_FORCE_DIST=1 python3 bench.py --workload g --steps 3 >> $O/bench_lines.jsonl 2>> $O/bench_lines.err"
python3 - <<'PY'
import json
for l in open("gpurun_out/r03/bench_lines.jsonl"):
    if not l.startswith("{"): continue
    j=json.loads(l); r=j["roofline"]
    print(j["config"]["workload"][:64], "| value %.0f ms/step %.3f kernel_ms %.3f frac %s %s %s" % (j["value"], j["ms_per_step"], r.get("kernel_ms",0), ("%.3f" % r["frac"]) if r.get("frac") else "-", j.get("invalid",""), "dist" if j.get("dist") else ""))
PY
