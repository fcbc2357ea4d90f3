#!/usr/bin/env python3
"""Fixtures for the hw3 affine score pass, generated from the UNMODIFIED hw3.cpp (oracle/_ref/libhw3_ref.so).
Dev container only.  Data files copied verbatim: the reference's small bundled inputs and its known-answer
output.phy (Multiple_Sequence_Alignment/, README.txt:31)."""
import json
import os
import random
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

REF = "/root/reference/Multiple_Sequence_Alignment"
SCORINGS = [(5, -4, -16, -4), (1, -1, -2, -1), (2, -3, -5, -2), (1, -1, 0, -1), (3, -2, -1, -3), (1, 1, 1, 1), (0, 0, 0, 0),
            (4, -5, 2, -1)]


def L(b):
    return b.decode("latin-1")


def main():
    assert O.have_ref3(), "run `make -C oracle ref` first"
    rng = random.Random(303)
    out = {}
    for f in ("input.fasta", "input16100.fasta", "input41000.fasta", "input161000.fasta", "output.phy"):
        shutil.copyfile(os.path.join(REF, f), os.path.join(HERE, "hw3_" + f))
    # all-pairs tables on the bundled inputs (upper triangle, i<j, row = sequence i: hw3.cpp:232-241)
    files = {}
    for f in ("input.fasta", "input16100.fasta", "input41000.fasta", "input161000.fasta"):
        recs = O.read_fasta_hw3(os.path.join(REF, f))
        seqs = [s for _, s in recs]
        per = {}
        for sc in [(5, -4, -16, -4), (1, -1, -2, -1), (2, -3, -5, -2)]:
            tab = [O.ref_affine_score(seqs[i], seqs[j], *sc) for i in range(len(seqs)) for j in range(i + 1, len(seqs))]
            c, sums = O.center(tab, len(seqs))
            per[",".join(map(str, sc))] = {"scores": tab, "center": c, "star": sums}
        files[f] = per
    out["bundled"] = files
    rnd = []
    for it in range(500):
        alpha = rng.choice([b"ACGT", b"AC", bytes(range(65, 91)), b"ACGTN"])
        a = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 80)))
        b = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 80)))
        if rng.random() < 0.4 and a:
            t = bytearray(a)
            for _ in range(rng.randint(0, 6)):
                if t:
                    pos, op = rng.randrange(len(t)), rng.random()
                    if op < 0.3:
                        del t[pos:pos + rng.randint(1, 4)]
                    elif op < 0.6:
                        t[pos:pos] = bytes(rng.choice(alpha) for _ in range(rng.randint(1, 4)))
                    else:
                        t[pos] = rng.choice(alpha)
            b = bytes(t)
        sc = rng.choice(SCORINGS)
        rnd.append(dict(a=L(a), b=L(b), scoring=list(sc), score=O.ref_affine_score(a, b, *sc)))
    for (n, m) in [(63, 64), (64, 64), (65, 130), (128, 129), (1, 300), (300, 1), (200, 257), (513, 70)]:
        a, b = O.gen(7, 0, n, n), O.gen(7, 1, m, m)
        for sc in [(5, -4, -16, -4), (1, -1, -2, -1)]:
            rnd.append(dict(a=L(a), b=L(b), scoring=list(sc), score=O.ref_affine_score(a, b, *sc)))
    out["random"] = rnd
    # generator KATs (SURVEY.md 8d streams): 16 x 1000 all pairs, and one 2000 x 3000 pair
    seqs = [O.gen(1, 2, i, 1000) for i in range(16)]
    tab = [O.ref_affine_score(seqs[i], seqs[j], 5, -4, -16, -4) for i in range(16) for j in range(i + 1, 16)]
    out["gen16x1000"] = {"scoring": [5, -4, -16, -4], "scores": tab, "sum": sum(tab)}
    out["gen_2000x3000"] = {"scoring": [5, -4, -16, -4],
                            "score": O.ref_affine_score(O.gen(1, 0, 0, 2000), O.gen(1, 1, 0, 3000), 5, -4, -16, -4)}
    # alignments with traceback (hw3.cpp:23-135, strings requested): the two gapped strings of the reference
    al = []
    for it in range(220):
        alpha = rng.choice([b"ACGT", b"AC", bytes(range(65, 91)), b"ACGTN"])
        a = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 70)))
        b = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 70)))
        if rng.random() < 0.5 and a:
            t = bytearray(a)
            for _ in range(rng.randint(0, 5)):
                if t:
                    pos, op = rng.randrange(len(t)), rng.random()
                    if op < 0.35:
                        del t[pos:pos + rng.randint(1, 5)]
                    elif op < 0.7:
                        t[pos:pos] = bytes(rng.choice(alpha) for _ in range(rng.randint(1, 5)))
                    else:
                        t[pos] = rng.choice(alpha)
            b = bytes(t)
        sc = rng.choice(SCORINGS)
        r = O.ref_affine_align(a, b, *sc)
        assert O.affine_align(a, b, *sc)["a1"] == r["a1"] and O.affine_align(a, b, *sc)["a2"] == r["a2"], (a, b, sc)
        al.append(dict(a=L(a), b=L(b), scoring=list(sc), score=r["score"], a1=L(r["a1"]), a2=L(r["a2"])))
    out["alignments"] = al
    with open(os.path.join(HERE, "hw3_affine.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("hw3_affine.json", os.path.getsize(os.path.join(HERE, "hw3_affine.json")))


if __name__ == "__main__":
    main()
