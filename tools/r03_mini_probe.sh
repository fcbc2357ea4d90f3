#!/bin/bash
# r03: what bounds the mini-stripe fill?  (1) fill time against the number of pairs (waves): flat = a wave's own pace, growing = a
# shared resource; (2) the same with the band stores compiled out (build/libpwalign_A.so, -DPWA_MINI_NOSTORE: timing only, results
# are wrong and the line says invalid).
source tools/gpu_steps.sh
O=gpurun_out/r03
mkdir -p $O
for lib in "" "$PWD/bioinformatics-algorithms_amd/csrc/build/libpwalign_A.so"; do
  for n in 256 1024 2048 4096 8192 16384; do
    for w in g gb; do
      if [ $w = gb ] && [ $n -gt 4096 ]; then continue; fi
      export PWA_LIB=$lib
      step ${w}_$n 120 python3 bench.py --workload $w --pairs $n --steps 3 --warmup 1 > $O/probe.json 2> $O/probe.err
      python3 - <<PY
import json
l = json.loads([x for x in open("$O/probe.json").read().splitlines() if x.startswith("{")][-1])
r = l["roofline"]
print("%-8s %-3s pairs %6d  fill %8.3f ms  walk %7.3f ms  written %6.2f GB -> %5.2f TB/s  kernel GCUPS %7.1f  %s" % ("nostore" if "$lib" else "product", "$w", $n, r["kernel_ms"], r["traceback_ms"], r["written_bytes_per_step"] / 1e9, r["written_bytes_per_step"] / r["kernel_ms"] / 1e9, r["kernel_gcups"], l.get("invalid", "")))
PY
    done
  done
done
