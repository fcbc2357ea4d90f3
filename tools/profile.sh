#!/bin/bash
# rocprofv3 evidence for one bench workload (run on the GPU box from the repo root):
#   tools/profile.sh <workload> <round-tag>
# pass 1: kernel trace + stats; passes 2-4: PMC counters, each in its own run (never mixed with
# trace domains).  Raw output goes to gpurun_out/prof_<workload>/; the summary that is meant to be
# committed is written by tools/summarize_prof.py into gpurun_out/profiles/ (copy it to profiles/).
W=${1:-c3}
TAG=${2:-r01}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$W
rm -rf $OUT; mkdir -p $OUT $PWD/gpurun_out/profiles
ARGS="bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
python3 tools/summarize_prof.py $OUT $W $TAG > $PWD/gpurun_out/profiles/${TAG}_${W}_rocprof_summary.md
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $PWD/gpurun_out/profiles/${TAG}_${W}_kernel_stats.csv
cat $PWD/gpurun_out/profiles/${TAG}_${W}_rocprof_summary.md
