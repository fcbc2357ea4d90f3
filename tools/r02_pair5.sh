#!/bin/bash
source tools/gpu_steps.sh
O=gpurun_out/r02
mkdir -p $O
step pytest_pair 900 bash -c "python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=5 -k 'align or overlap or matrices or pipeline or large_pair or c5 or cli or dropin or plain or beyond or degenerate' > $O/gputest_pair.log 2>&1"
tail -8 $O/gputest_pair.log
for rep in 1 2; do
for v in shift noshift; do
  for w in c5 g; do
    if [ $v = noshift ]; then export PWA_NO_GAP_SHIFT=1; else unset PWA_NO_GAP_SHIFT; fi
    step ${v}_${w} 120 bash -c "python bench.py --workload $w --steps 8 --warmup 2 > $O/gs_${v}_${w}.json 2> $O/gs.err"
    python - "$O/gs_${v}_${w}.json" $v $w <<'PY'
import json,sys
l=json.load(open(sys.argv[1])); r=l["roofline"]
print(sys.argv[2], sys.argv[3], "fill_ms %.3f tb_ms %.3f ms/step %.2f" % (r["kernel_ms"], r.get("traceback_ms",0), l["ms_per_step"]), l.get("result"))
PY
  done
done
done
unset PWA_NO_GAP_SHIFT
step scal_rl4 200 bash -c "PWA_FORCE_RL=4 python tools/pair_scaling.py nw > $O/scaling_rl4.txt 2>&1"
grep -v "m  20000" $O/scaling_rl4.txt
