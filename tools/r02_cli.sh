#!/bin/bash
# hw2_amd at scale (262144 index-paired pairs 150 x 2000, 569 MB of FASTA): wall of -l and -g, with the library's own marks
source tools/gpu_steps.sh
O=gpurun_out/r02
mkdir -p $O /tmp/scale_cli
D=/tmp/scale_cli
step gen 600 python3 - <<'PY'
import sys
sys.path.insert(0, ".")
import bench
n = 262144
with open("/tmp/scale_cli/p.fasta", "wb") as f:
    for i in range(n): f.write(b">p%d\n" % i + bench.gen(1, 0, i, 150) + b"\n")
with open("/tmp/scale_cli/t.fasta", "wb") as f:
    for i in range(n): f.write(b">t%d\n" % i + bench.gen(1, 1, i, 2000) + b"\n")
PY
ls -la $D/*.fasta
EXE=bioinformatics-algorithms_amd/host/hw2_amd
for rep in 1 2 3; do
  for f in l g; do
    /usr/bin/env bash -c "time $EXE -$f -p $D/p.fasta -t $D/t.fasta -o $D/$f.txt -s 1 -1 -1" 2>&1 | grep real | sed "s/^/-$f /"
  done
done
PWA_DEBUG=1 $EXE -l -p $D/p.fasta -t $D/t.fasta -o $D/l2.txt -s 1 -1 -1 2>&1 | tail -12
PWA_DEBUG=1 $EXE -g -p $D/p.fasta -t $D/t.fasta -o $D/g2.txt -s 1 -1 -1 2>&1 | grep -v "chunk\|task list\|fill launch\|fill + walk\|results\|scatter" | tail -8
PWA_DEBUG=1 $EXE -g -p $D/p.fasta -t $D/t.fasta -o $D/g2.txt -s 1 -1 -1 2>&1 | awk '/fill \+ walk/{d+=$(NF-1)} /chunk desc/{c+=$(NF-1)} /task list/{t+=$(NF-1)} /results/{r+=$(NF-1)} /scatter/{s+=$(NF-1)} END{print "sum over chunks: device", d, "desc", c, "tasks", t, "d2h", r, "scatter", s}'
cut -c1-80 $D/l.txt | head -4; cut -c1-80 $D/g.txt | head -4
sha256sum $D/l.txt $D/g.txt
