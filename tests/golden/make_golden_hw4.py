#!/usr/bin/env python3
"""Fixtures for hw4 (NW distance with tie-break diag >= up >= left, UPGMA, Newick) from the UNMODIFIED hw4.cpp
(oracle/_ref).  Dev container only.  input.fasta / tree.txt are the reference's own data + known answer."""
import json
import os
import random
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

REF = "/root/reference/hw4"


def L(b):
    return b.decode("latin-1")


def main():
    assert O.have_ref4()
    rng = random.Random(404)
    shutil.copyfile(os.path.join(REF, "input.fasta"), os.path.join(HERE, "hw4_input.fasta"))
    shutil.copyfile(os.path.join(REF, "tree.txt"), os.path.join(HERE, "hw4_tree.txt"))
    out = {"pairs": [], "cli": []}
    scorings = [(1, -1, -1), (2, -3, -5), (5, -4, -4), (1, -3, -1), (1, 1, 1), (0, 0, 0), (-1, 2, 1), (1, -1, 0)]
    for it in range(500):
        alpha = rng.choice([b"ACGT", b"AC", bytes(range(65, 91))])
        a = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 90)))
        b = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 90)))
        if rng.random() < 0.4 and a:
            t = bytearray(a)
            for _ in range(rng.randint(0, 6)):
                if t:
                    pos, op = rng.randrange(len(t)), rng.random()
                    if op < 0.3:
                        del t[pos]
                    elif op < 0.6:
                        t.insert(pos, rng.choice(alpha))
                    else:
                        t[pos] = rng.choice(alpha)
            b = bytes(t)
        sc = rng.choice(scorings)
        out["pairs"].append(dict(a=L(a), b=L(b), scoring=list(sc), dist=O.ref_nw_distance(a, b, *sc)))
    for (n, m) in [(63, 64), (64, 64), (65, 130), (96, 97), (128, 129), (1, 300), (300, 1), (200, 257), (513, 70)]:
        a, b = O.gen(7, 0, n, n), O.gen(7, 1, m, m)
        for sc in [(1, -1, -1), (2, -3, -5)]:
            out["pairs"].append(dict(a=L(a), b=L(b), scoring=list(sc), dist=O.ref_nw_distance(a, b, *sc)))
    seqs = [O.gen(1, 2, i, 1000) for i in range(16)]
    tab = [O.ref_nw_distance(seqs[i], seqs[j], 1, -1, -1) for i in range(16) for j in range(i + 1, 16)]
    out["gen16x1000"] = {"scoring": [1, -1, -1], "dist": tab}
    # CLI: random FASTA files -> tree
    with tempfile.TemporaryDirectory() as td:
        for case in range(12):
            n = rng.randint(2, 9)
            recs = []
            base = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(20, 120)))
            for i in range(n):
                t = bytearray(base)
                for _ in range(rng.randint(0, 12)):
                    pos = rng.randrange(len(t))
                    r = rng.random()
                    if r < 0.3 and len(t) > 5:
                        del t[pos]
                    elif r < 0.6:
                        t.insert(pos, rng.choice(b"ACGT"))
                    else:
                        t[pos] = rng.choice(b"ACGT")
                recs.append((b"sp%d|x%d" % (case, i), bytes(t)))
            text = b""
            for i, (h, s) in enumerate(recs):
                half = len(s) // 2
                eol = b"\r\n" if case % 3 == 0 else b"\n"
                text += b">" + h + eol + s[:half] + eol + (b"\n" if case % 4 == 1 else b"") + s[half:] + eol
            sc = scorings[case % 4]
            open(os.path.join(td, "in.fa"), "wb").write(text)
            pr = subprocess.run([O.REF4_CLI, "-i", "in.fa", "-t", "tree.txt", "-s"] + [str(x) for x in sc], cwd=td,
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            out["cli"].append(dict(fasta=L(text), scoring=list(sc), rc=pr.returncode, stderr=L(pr.stderr),
                                   tree=L(open(os.path.join(td, "tree.txt"), "rb").read())))
        for args in (["-i", "missing.fa", "-t", "tree.txt", "-s", "1", "-1", "-1"], ["-i", "in.fa"], ["-x", "in.fa", "-t", "t", "-s", "1", "-1", "-1"]):
            pr = subprocess.run([O.REF4_CLI] + args, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            out["cli"].append(dict(args=args, rc=pr.returncode, stderr=L(pr.stderr.replace(O.REF4_CLI.encode(), b"hw4"))))
    json.dump(out, open(os.path.join(HERE, "hw4.json"), "w"), indent=0)
    print("hw4.json", os.path.getsize(os.path.join(HERE, "hw4.json")))


if __name__ == "__main__":
    main()
