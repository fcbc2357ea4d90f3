#!/bin/bash
# hw2_amd at scale: 262144 index-paired pairs (150 x 2000), -l (per-lane-text strips) and -g (pair engine, chunked bands).
set -e
D=/tmp/scale_cli; mkdir -p $D
python3 - <<'PY'
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
time $EXE -l -p $D/p.fasta -t $D/t.fasta -o $D/l.txt -s 1 -1 -1
export PWA_FORCE_LANES=0; time $EXE -l -p $D/p.fasta -t $D/t.fasta -o $D/l_grouped.txt -s 1 -1 -1; unset PWA_FORCE_LANES
cmp $D/l.txt $D/l_grouped.txt && echo "-l: per-lane-text and text-grouped forms agree"
cut -c1-100 $D/l.txt
time $EXE -g -p $D/p.fasta -t $D/t.fasta -o $D/g.txt -s 1 -1 -1
cut -c1-100 $D/g.txt
