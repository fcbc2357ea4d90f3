#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r02
mkdir -p $O
step scal_rl4 200 bash -c "PWA_FORCE_RL=4 python tools/pair_scaling.py nw > $O/scaling_rl4.txt 2>&1"
step scal_rl4_nopad 200 bash -c "PWA_NO_LDS_PAD=1 PWA_FORCE_RL=4 python tools/pair_scaling.py nw > $O/scaling_rl4_nopad.txt 2>&1"
step scal_rl4_w1 200 bash -c "PWA_FORCE_W=1 PWA_FORCE_RL=4 python tools/pair_scaling.py nw > $O/scaling_rl4_w1.txt 2>&1"
step scal_rl2 200 bash -c "PWA_FORCE_RL=2 python tools/pair_scaling.py nw > $O/scaling_rl2.txt 2>&1"
cat $O/scaling_rl4.txt $O/scaling_rl4_nopad.txt $O/scaling_rl4_w1.txt $O/scaling_rl2.txt | grep -v "m  20000"
for w in c2 c5; do
  step bench_$w 120 bash -c "python bench.py --workload $w --steps 10 --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err"
done
for f in $O/bench_c2.json $O/bench_c5.json; do python - "$f" <<'PY'
import json,sys
try:
    l=json.load(open(sys.argv[1])); r=l["roofline"]
    print(sys.argv[1].split("/")[-1], "value %.1f ms/step %.2f fill_ms %.3f tb_ms %.3f frac %.3f" % (l["value"], l["ms_per_step"], r["kernel_ms"], r.get("traceback_ms",0), r["frac"]), l.get("result"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
