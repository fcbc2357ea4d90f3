#!/bin/bash
# r03: hw2_amd at scale, one-shot score calls pipelined over runs (default) against one run per arena (PWA_NO_PIPELINE=1):
#   A  262144 index-paired pairs 150 x 2000 (569 MB of FASTA)      B  294912 pairs 150 x 15000 (4.5 GB: more than one 4 GiB arena)
source tools/gpu_steps.sh
D=/tmp/scale_cli
mkdir -p $D
EXE=bioinformatics-algorithms_amd/host/hw2_amd
gen() {  # name n_pairs text_len
python3 - "$1" "$2" "$3" <<'PY'
import sys
sys.path.insert(0, ".")
import bench
name, n, tl = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
with open("/tmp/scale_cli/p%s.fasta" % name, "wb") as f:
    for i in range(n): f.write(b">p%d\n" % i + bench.gen(1, 0, i, 150) + b"\n")
with open("/tmp/scale_cli/t%s.fasta" % name, "wb") as f:
    for i in range(n): f.write(b">t%d\n" % i + bench.gen(1, 1, i, tl) + b"\n")
PY
}
run() {  # name flags...
  local name=$1
  for mode in pipelined single-run; do
    if [ $mode = single-run ]; then export PWA_NO_PIPELINE=1; else unset PWA_NO_PIPELINE; fi
    for f in l g; do
      for rep in 1 2 3; do
        /usr/bin/env bash -c "time $EXE -$f -p $D/p$name.fasta -t $D/t$name.fasta -o $D/$f$name.$mode.txt -s 1 -1 -1" 2>&1 | grep real | sed "s/^/input $name  -$f  $mode  /"
      done
    done
    PWA_DEBUG=1 $EXE -l -p $D/p$name.fasta -t $D/t$name.fasta -o $D/dbg.txt -s 1 -1 -1 2>&1 | grep "scores pass" | sed "s/^/input $name  -l  $mode  /"
  done
  unset PWA_NO_PIPELINE
  sha256sum $D/l$name.pipelined.txt $D/l$name.single-run.txt $D/g$name.pipelined.txt $D/g$name.single-run.txt | awk '{print substr($1,1,16), $2}'
}
step genA 600 gen A 262144 2000
ls -la $D/*A.fasta | awk '{print $5, $9}'
run A
step genB 900 gen B 294912 15000
ls -la $D/*B.fasta | awk '{print $5, $9}'
run B
rm -f $D/*B.fasta
