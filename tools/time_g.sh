#!/bin/bash
# Where the wall time of the `-g` shape goes (4096 full global alignments 150 x 10000 through the hw2-compatible CLI).
set -e
D=/tmp/time_g; mkdir -p $D gpurun_out
python3 - <<'PY'
import sys
sys.path.insert(0, ".")
import bench
n = 4096
with open("/tmp/time_g/p.fasta", "w") as f:
    for i in range(n): f.write(">p%d\n%s\n" % (i, bench.gen(1, 0, i, 150).decode()))
with open("/tmp/time_g/t.fasta", "w") as f:
    for i in range(n): f.write(">t%d\n%s\n" % (i, bench.gen(1, 1, i % 256, 10000).decode()))
PY
EXE=bioinformatics-algorithms_amd/host/hw2_amd
for i in 1 2; do
  export PWA_DEBUG=1; time $EXE -g -p $D/p.fasta -t $D/t.fasta -o $D/g.txt -s 1 -1 -1; unset PWA_DEBUG
done
cat $D/g.txt | cut -c1-120
time $EXE -l -p $D/p.fasta -t $D/t.fasta -o $D/l.txt -s 1 -1 -1
cat $D/l.txt | cut -c1-120
