#!/usr/bin/env python3
"""gen_fasta_pairs.py <dir> <name> <n_pairs> <pattern_len> <text_len>: writes <dir>/p<name>.fasta and <dir>/t<name>.fasta, n_pairs records each, from the
SURVEY 8(d) generator (streams 0 and 1), for the hw2_amd scale runs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402

d, name, n, pl, tl = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
os.makedirs(d, exist_ok=True)
with open(os.path.join(d, "p%s.fasta" % name), "wb") as f:
    for i in range(n):
        f.write(b">p%d\n" % i + bench.gen(1, 0, i, pl) + b"\n")
with open(os.path.join(d, "t%s.fasta" % name), "wb") as f:
    for i in range(n):
        f.write(b">t%d\n" % i + bench.gen(1, 1, i, tl) + b"\n")
