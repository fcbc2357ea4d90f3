#!/bin/bash
# A/B on ONE box: build/libpwalign_A.so (PWA_LIB) against the in-tree library, alternating, C5 and C2 fills
source tools/gpu_steps.sh
O=gpurun_out/r02
mkdir -p $O
for rep in 1 2 3; do
  for v in A B; do
    for w in c5 c2; do
      if [ $v = A ]; then export PWA_LIB=$PWD/build/libpwalign_A.so; else unset PWA_LIB; fi
      step ab_${v}_${w}_$rep 120 bash -c "python bench.py --workload $w --steps 8 --warmup 2 > $O/ab_${v}_${w}_$rep.json 2> $O/ab.err"
      python - "$O/ab_${v}_${w}_$rep.json" $v $w <<'PY'
import json,sys
l=json.load(open(sys.argv[1])); r=l["roofline"]
print(sys.argv[2], sys.argv[3], "fill_ms %.3f tb_ms %.3f" % (r["kernel_ms"], r.get("traceback_ms",0)))
PY
    done
  done
done
