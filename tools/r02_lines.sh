#!/bin/bash
# every bench workload once, lines collected in gpurun_out/r02/bench_lines.jsonl (copied to profiles/r02_bench_lines.jsonl)
source tools/gpu_steps.sh
O=gpurun_out/r02
mkdir -p $O
: > $O/bench_lines.jsonl
step c3 400 bash -c "python bench.py >> $O/bench_lines.jsonl 2>> $O/bench_lines.err"
for w in c3i c4 hw3 hw4; do
  step $w 400 bash -c "python bench.py --workload $w >> $O/bench_lines.jsonl 2>> $O/bench_lines.err"
done
step c3i_nw 400 bash -c "python bench.py --workload c3i --nw >> $O/bench_lines.jsonl 2>> $O/bench_lines.err"
for w in c2 c2b c5 g gb; do
  step $w 200 bash -c "python bench.py --workload $w --steps 10 --warmup 2 >> $O/bench_lines.jsonl 2>> $O/bench_lines.err"
done
step gb128 200 bash -c "python bench.py --workload gb --plen 128 --steps 10 --warmup 2 >> $O/bench_lines.jsonl 2>> $O/bench_lines.err"
step dist1 200 bash -c "BENCH_FORCE_DIST=1 python bench.py --steps 3 --no-cpu-baseline >> $O/bench_lines.jsonl 2>> $O/bench_lines.err"
python - <<'PY'
import json
for l in open("gpurun_out/r02/bench_lines.jsonl"):
    j=json.loads(l); r=j["roofline"]
    print(j["config"]["workload"][:58], "| value %.0f ms/step %.2f kernel_ms %.3f frac %.3f %s %s" % (j["value"], j["ms_per_step"], r.get("kernel_ms",0), r.get("frac") or 0, j.get("invalid",""), j.get("dist","")))
PY
