#!/bin/bash
# Stall counters for the C3 kernel in its two forms (LDS-paired strips vs HBM hand-off), one PMC pass per
# counter group and per form; sums per kernel written to gpurun_out/pmc_compare.txt.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_cmp
rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --workload c3 --steps 2 --warmup 1 --no-cpu-baseline"
G1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
G2="SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU"
G3="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_BRANCH"
for form in 1 0; do
  export PWA_PAIRED=$form
  i=0
  for G in "$G1" "$G2" "$G3"; do
    i=$((i+1))
    rocprofv3 --pmc $G --output-format csv -d $OUT/f${form}_g$i -- python3 $ARGS > $OUT/f${form}_g$i.log 2>&1 || echo "pass f$form g$i failed"
  done
done
python3 - <<'PY' > $PWD/gpurun_out/pmc_compare.txt
import csv, glob, collections
for form in (1, 0):
    tot = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(f"gpurun_out/pmc_cmp/f{form}_g*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "batch_scores" not in r["Kernel_Name"]: continue
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print("paired" if form else "unpaired")
    for k in sorted(tot): print(f"  {k:24s} {tot[k]/max(n[k],1):.4g}   (mean of {n[k]} dispatches)")
PY
cat $PWD/gpurun_out/pmc_compare.txt
