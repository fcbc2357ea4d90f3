#!/bin/bash
# hw2_amd wall on 262144 pairs 150 x 2000 (569 MB of FASTA), five runs per mode
source tools/gpu_steps.sh
D=/tmp/scale_cli
mkdir -p $D
EXE=bioinformatics-algorithms_amd/host/hw2_amd
step genA 600 python3 tools/gen_fasta_pairs.py $D A 262144 150 2000
for f in l g; do
  for rep in 1 2 3 4 5; do
    /usr/bin/env bash -c "time $EXE -$f -p $D/pA.fasta -t $D/tA.fasta -o $D/$f.txt -s 1 -1 -1" 2>&1 | grep real | sed "s/^/-$f  /"
  done
done
sha256sum $D/l.txt $D/g.txt | awk '{print substr($1,1,16), $2}'
