#!/bin/bash
# hw3_amd against the compiled reference program on a family of 64 (and 256) sequences of ~1000 bp.
D=/tmp/time_hw3; mkdir -p $D
python3 - <<'PY'
import random
rng = random.Random(5)
base = [rng.choice("ACGT") for _ in range(1000)]
def mut(s, rate=0.1):
    out = []
    for c in s:
        r = rng.random()
        if r < rate / 3: continue
        if r < 2 * rate / 3: out.append(rng.choice("ACGT"))
        out.append(rng.choice("ACGT") if r > 1 - rate / 3 else c)
    return "".join(out)
for n in (64, 256):
    with open("/tmp/time_hw3/fam%d.fasta" % n, "w") as f:
        for i in range(n):
            f.write(">seq%04d\n%s\n" % (i, mut(base)))
PY
EXE=bioinformatics-algorithms_amd/host/hw3_amd
time $EXE -i $D/fam64.fasta -o $D/a64.phy -s 5:-4:-16:-4
time $EXE -i $D/fam256.fasta -o $D/a256.phy -s 5:-4:-16:-4
if [ -x oracle/_ref/hw3_ref ]; then
  time oracle/_ref/hw3_ref -i $D/fam64.fasta -o $D/r64.phy -s 5:-4:-16:-4
  cmp $D/a64.phy $D/r64.phy && echo "64 sequences: identical to the reference program's output"
fi
head -c 300 $D/a256.phy; echo
