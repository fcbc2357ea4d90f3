#!/bin/bash
# r03: hw2_amd at scale.  The runs of a one-shot score call are pipelined; three forms: the default (one run per 4 GiB arena),
# six runs on purpose (PWA_PIPE_RUNS=6: first kernels start when a sixth of the input is up), strictly serial runs (PWA_NO_PIPELINE=1):
#   A  262144 index-paired pairs 150 x 2000 (569 MB of FASTA)      B  294912 pairs 150 x 15000 (4.5 GB: more than one 4 GiB arena)
source tools/gpu_steps.sh
D=/tmp/scale_cli
mkdir -p $D
EXE=bioinformatics-algorithms_amd/host/hw2_amd
run() {  # name flags...
  local name=$1
  for mode in default six-runs serial; do
    unset PWA_NO_PIPELINE PWA_PIPE_RUNS
    if [ $mode = serial ]; then export PWA_NO_PIPELINE=1; fi
    if [ $mode = six-runs ]; then export PWA_PIPE_RUNS=6; fi
    for f in l g; do
      for rep in 1 2 3; do
        /usr/bin/env bash -c "time $EXE -$f -p $D/p$name.fasta -t $D/t$name.fasta -o $D/$f$name.$mode.txt -s 1 -1 -1" 2>&1 | grep real | sed "s/^/input $name  -$f  $mode  /"
      done
    done
    PWA_DEBUG=1 $EXE -l -p $D/p$name.fasta -t $D/t$name.fasta -o $D/dbg.txt -s 1 -1 -1 2>&1 | grep "scores pass" | sed "s/^/input $name  -l  $mode  /"
    if [ $mode = default ]; then PWA_DEBUG=1 $EXE -g -p $D/p$name.fasta -t $D/t$name.fasta -o $D/dbg.txt -s 1 -1 -1 2>&1 | grep "overlaps:" | sed "s/^/input $name  -g  /"; fi
  done
  unset PWA_NO_PIPELINE PWA_PIPE_RUNS
  sha256sum $D/l$name.default.txt $D/l$name.six-runs.txt $D/l$name.serial.txt $D/g$name.default.txt | awk '{print substr($1,1,16), $2}'
}
step genA 600 python3 tools/gen_fasta_pairs.py $D A 262144 150 2000
ls -la $D/*A.fasta | awk '{print $5, $9}'
run A
step genB 900 python3 tools/gen_fasta_pairs.py $D B 294912 150 15000
ls -la $D/*B.fasta | awk '{print $5, $9}'
run B
rm -f $D/*B.fasta
