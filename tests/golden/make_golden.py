#!/usr/bin/env python3
"""Generate tests/golden/*.json from the UNMODIFIED reference (oracle/_ref, built by `make -C oracle ref`
from /root/reference/Local_Global_Alignment/hw2.cpp).  Dev container only: the reference does not
travel to the GPU box, these small fixtures do.

Fixtures are DATA: inputs and the reference's outputs for them (long strings as sha256).
The four small data files patterns.fasta / texts.fasta / global.txt / local.txt are the reference's own
bundled input and known-answer output (Local_Global_Alignment/, README.txt:16) and are copied verbatim.

Usage:  python tests/golden/make_golden.py
"""
import hashlib
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

REF_DATA = "/root/reference/Local_Global_Alignment"
SCORINGS = [(1, -1, -1), (2, -3, -5), (5, -4, -4), (1, -3, -1), (0, 0, 0), (1, 1, 1), (-1, 2, 1), (3, -1, 2),
            (1, -1, 0), (2, -1, -3)]


def sha(b):
    return hashlib.sha256(b).hexdigest()


def L(b):
    return b.decode("latin-1")


def full_record(mode, p, t, sc):
    r = O.ref_align(mode, p, t, *sc)
    return dict(mode=mode, p=L(p), t=L(t), scoring=list(sc), score=r["score"], cigar=L(r["cigar"]), mdz=L(r["mdz"]),
                aligned_pattern=L(r["aligned_pattern"]), aligned_reference=L(r["aligned_reference"]),
                overlap=r["overlap"])


def hashed_record(mode, gen_p, gen_t, sc):
    p = O.gen(*gen_p)
    t = O.gen(*gen_t)
    r = O.ref_align(mode, p, t, *sc)
    return dict(mode=mode, gen_p=list(gen_p), gen_t=list(gen_t), scoring=list(sc), score=r["score"],
                overlap=r["overlap"], aligned_len=len(r["aligned_pattern"]), cigar_sha256=sha(r["cigar"]),
                mdz_sha256=sha(r["mdz"]), aligned_pattern_sha256=sha(r["aligned_pattern"]),
                aligned_reference_sha256=sha(r["aligned_reference"]), cigar_head=L(r["cigar"][:64]))


def read_fasta_py(path):
    seqs, cur = [], b""
    for line in open(path, "rb").read().split(b"\n"):
        line = line.rstrip(b" \t\r\n\v\f")
        if not line:
            continue
        if line[:1] == b">":
            if cur:
                seqs.append(cur)
                cur = b""
        else:
            cur += line
    if cur:
        seqs.append(cur)
    return seqs


def main():
    assert O.have_ref(), "run `make -C oracle ref` first (needs /root/reference)"
    rng = random.Random(481)

    # --- the reference's own bundled data + known-answer outputs (copied as data)
    for f in ("patterns.fasta", "texts.fasta", "global.txt", "local.txt"):
        shutil.copyfile(os.path.join(REF_DATA, f), os.path.join(HERE, f))

    out = {}
    # --- bundled pairs under every scoring
    pats = read_fasta_py(os.path.join(REF_DATA, "patterns.fasta"))
    txts = read_fasta_py(os.path.join(REF_DATA, "texts.fasta"))
    out["bundled"] = [full_record(m, p, t, sc) for sc in SCORINGS for (p, t) in zip(pats, txts) for m in ("nw", "sw")]

    # --- edge cases of SURVEY.md section 4
    edge_pairs = [(b"AAAA", b"CCCC"), (b"A", b"A"), (b"A", b"C"), (b"ACGT", b"acgt"), (b"", b"ACGT"), (b"ACGT", b""),
                  (b"", b""), (b"AAAAAAAA", b"AAAA"), (b"AAAA", b"AAAAAAAA"), (b"ACGTACGT", b"ACGTACGT"),
                  (b"GATTACA", b"GCATGCU"), (b"N" * 70, b"N" * 65), (b"AC" * 40, b"CA" * 40),
                  (bytes(range(1, 128)), bytes(range(127, 0, -1)))]
    out["edge"] = [full_record(m, p, t, sc) for (p, t) in edge_pairs for sc in [(1, -1, -1), (1, -3, -1), (2, -3, -5), (1, 1, 1)]
                   for m in ("nw", "sw")]

    # --- seeded random small cases: iid and related pairs, several alphabets, tie-heavy scorings
    rnd = []
    for it in range(400):
        alpha = rng.choice([b"ACGT", b"AC", b"ACGTN", bytes(range(65, 91))])
        n1, n2 = rng.randint(1, 90), rng.randint(1, 90)
        p = bytes(rng.choice(alpha) for _ in range(n1))
        t = bytes(rng.choice(alpha) for _ in range(n2))
        if rng.random() < 0.4:
            tt = bytearray(p)
            for _ in range(rng.randint(0, 8)):
                if not tt:
                    break
                pos, op = rng.randrange(len(tt)), rng.random()
                if op < 0.3:
                    del tt[pos]
                elif op < 0.6:
                    tt.insert(pos, rng.choice(alpha))
                else:
                    tt[pos] = rng.choice(alpha)
            t = bytes(tt) or b"A"
        sc = rng.choice(SCORINGS)
        for m in ("nw", "sw"):
            rnd.append(full_record(m, p, t, sc))
    # sizes that straddle the kernels' stripe / strip / chunk edges
    for (n1, n2) in [(63, 64), (64, 64), (65, 64), (64, 65), (127, 130), (128, 128), (129, 257), (150, 300), (1, 300),
                     (300, 1), (256, 256), (257, 255), (192, 200), (513, 70), (70, 513)]:
        p = O.gen(7, 0, n1, n1)
        t = O.gen(7, 1, n2, n2)
        for sc in [(1, -1, -1), (2, -3, -5)]:
            for m in ("nw", "sw"):
                rnd.append(full_record(m, p, t, sc))
    out["random"] = rnd

    # --- sequences that contain a literal '-': overlapLongestExactMatch (hw2.cpp:269) treats such a column as a gap
    # column even though it came from the input, and prepareMDZString / the gapped strings carry the byte through
    dash = []
    dash_pairs = [(b"AC-GT", b"AC-GT"), (b"ACGA", b"ACGA"), (b"--------", b"--------"), (b"ACGT-ACGTACGT", b"ACGTACGT-ACGT"),
                  (b"A-C-G-T-", b"-A-C-G-T"), (b"AAAA-AAAA", b"AAAAAAAA"), (b"-", b"-"), (b"-ACGT", b"ACGT-"),
                  (b"ACGTACGTAC--GTACGTACGT", b"ACGTACGTAC--GTACGTACGT")]
    for it in range(60):
        alpha = rng.choice([b"ACGT-", b"A-", b"ACGTN-"])
        n1, n2 = rng.randint(1, 200), rng.randint(1, 200)
        p = bytes(rng.choice(alpha) for _ in range(n1))
        t = bytes(rng.choice(alpha) for _ in range(n2))
        if rng.random() < 0.6:   # related pair: long diagonal runs through '-' symbols
            tt = bytearray(p)
            for _ in range(rng.randint(0, 6)):
                pos = rng.randrange(len(tt))
                tt[pos] = rng.choice(alpha)
            t = bytes(tt)
        dash_pairs.append((p, t))
    for (p, t) in dash_pairs:
        for sc in [(1, -1, -1), (2, -3, -5), (1, 1, 1)]:
            for m in ("nw", "sw"):
                dash.append(full_record(m, p, t, sc))
    out["dash"] = dash

    # --- generator KATs of SURVEY.md 8(d) (long strings hashed)
    kat = []
    for (n, m, sc) in [(64, 64, (1, -1, -1)), (150, 10000, (1, -1, -1)), (150, 10000, (2, -3, -5)),
                       (1000, 1000, (1, -1, -1)), (1000, 1000, (5, -4, -4)), (2000, 3000, (1, -1, -1)),
                       (10000, 10000, (1, -1, -1))]:
        for mode in ("nw", "sw"):
            kat.append(hashed_record(mode, (1, 0, 0, n), (1, 1, 0, m), sc))
    out["kat"] = kat

    # --- scores x lengths beyond 2^28 (the engine's packed-key traceback form must hand over to its plain int32 form):
    # the reference's recurrences work for anything its `int` holds (hw2.cpp:140-153, 206-222)
    big = []
    for (n, m, sc) in [(1500, 1600, (100000, -100000, -100000)), (1400, 1300, (1 << 20, -(1 << 19), -3)),
                       (900, 2000, (7, -250000, -120000)), (3000, 200, (100000, -100000, 100000)),
                       (1600, 1500, (-100000, 100000, -50000))]:
        for mode in ("nw", "sw"):
            big.append(hashed_record(mode, (3, 0, n, n), (3, 1, m, m), sc))
    out["bigscore"] = big

    # --- the real 16 x ~1000 bp file of the sibling program, all pairs, both modes, two scorings
    real = read_fasta_py("/root/reference/Multiple_Sequence_Alignment/input161000.fasta")
    assert len(real) == 16
    c4r = {"file": "hw3_input161000.fasta", "n_seq": 16, "scorings": {}}
    for sc in [(1, -1, -1), (5, -4, -4)]:
        nw = [O.ref_align("nw", real[i], real[j], *sc)["score"] for i in range(16) for j in range(i + 1, 16)]
        sw = [O.ref_align("sw", real[i], real[j], *sc)["score"] for i in range(16) for j in range(i + 1, 16)]
        c4r["scorings"][",".join(map(str, sc))] = {"nw": nw, "sw": sw, "nw_sum": sum(nw), "nw_min": min(nw), "nw_max": max(nw),
                                                    "sw_sum": sum(sw)}
    assert c4r["scorings"]["1,-1,-1"]["nw_sum"] == 3013 and c4r["scorings"]["5,-4,-4"]["sw_sum"] == 102928
    out["c4_real"] = c4r

    # --- batched score tables (scores-only kernels): small C3-shaped and C4-shaped batches
    pats = [O.gen(1, 0, i, 150) for i in range(96)]
    txts = [O.gen(1, 1, i, 1000) for i in range(4)]
    c3 = {"n_patterns": 96, "pattern_len": 150, "n_texts": 4, "text_len": 1000, "scorings": {}}
    for sc in [(1, -1, -1), (2, -3, -5)]:
        tab = []
        for p in pats:
            for t in txts:
                s, ei, ej = O.score("sw", p, t, *sc)
                assert s == O.ref_align("sw", p, t, *sc)["score"]
                tab.append([s, ei, ej])
        c3["scorings"][",".join(map(str, sc))] = tab
    out["c3_small"] = c3
    seqs = [O.gen(1, 2, i, 1000) for i in range(16)]
    tot, tab = 0, []
    for i in range(16):
        for j in range(i + 1, 16):
            s = O.ref_align("nw", seqs[i], seqs[j], 1, -1, -1)["score"]
            tab.append(s)
            tot += s
    assert tot == 11397, tot
    out["c4_small"] = {"n_seq": 16, "len": 1000, "scoring": [1, -1, -1], "scores_upper_triangle": tab, "sum": tot}

    # --- CLI behaviour: rc, stderr text, output bytes (argv[0] normalised to 'hw2')
    cli = []
    with tempfile.TemporaryDirectory() as td:
        for f in ("patterns.fasta", "texts.fasta"):
            shutil.copyfile(os.path.join(REF_DATA, f), os.path.join(td, f))
        open(os.path.join(td, "two.fasta"), "w").write(">a\nACGT\n>b\nGGGG\n")
        open(os.path.join(td, "messy_p.fasta"), "wb").write(b">p1 \r\nACGT \r\nAC GT\t\r\n\r\n>empty\n>p2\nacgt\nACGT  \n>p3\nTTTT")
        open(os.path.join(td, "messy_t.fasta"), "wb").write(b"ACGTACGT\n>t2\n\nACGTACG\n>t3\nTTTAT\n")
        # the winner of -g changes when a '-' inside a sequence stops counting towards the overlap (hw2.cpp:269, 344)
        open(os.path.join(td, "dash.fasta"), "w").write(">a\nAC-GT\n>b\nACGA\n")
        open(os.path.join(td, "dash_p.fasta"), "w").write(">a\nACGT-ACGTACGT\n>b\nAC--GT\n>c\nTTTTT\n")
        open(os.path.join(td, "dash_t.fasta"), "w").write(">a\nACGTACGT-ACGT\n>b\nAC--GT\n>c\nTTTAT\n")
        # scores x lengths beyond 2^28 through the whole program (`hw2_amd -g -s 100000 ...` on 3 kb must not refuse)
        open(os.path.join(td, "big_p.fasta"), "wb").write(b">a\n" + O.gen(5, 0, 0, 1500) + b"\n>b\n" + O.gen(5, 0, 1, 1700) + b"\n")
        open(os.path.join(td, "big_t.fasta"), "wb").write(b">a\n" + O.gen(5, 1, 0, 1600) + b"\n>b\n" + O.gen(5, 1, 1, 1400) + b"\n")
        files = {f: L(open(os.path.join(td, f), "rb").read()) for f in ("two.fasta", "messy_p.fasta", "messy_t.fasta", "dash.fasta",
                                                                          "dash_p.fasta", "dash_t.fasta", "big_p.fasta", "big_t.fasta")}
        cases = [
            ["-g", "-p", "patterns.fasta", "-t", "texts.fasta", "-o", "out.txt", "-s", "1", "-1", "-1"],
            ["-l", "-p", "patterns.fasta", "-t", "texts.fasta", "-o", "out.txt", "-s", "1", "-1", "-1"],
            ["-g", "-l", "-p", "patterns.fasta", "-t", "texts.fasta", "-o", "out.txt", "-s", "1", "-1", "-1"],
            ["-x", "-p", "patterns.fasta", "-t", "texts.fasta", "-o", "out.txt", "-s", "1", "-1", "-1"],
            ["-l", "-p", "patterns.fasta", "-t", "texts.fasta", "-o", "out.txt", "-s", "2", "-3", "-5"],
            ["-g", "-p", "patterns.fasta", "-t", "texts.fasta", "-o", "out.txt", "-s", "5", "-4", "-4"],
            ["-l", "-p", "patterns.fasta", "-t", "texts.fasta", "-o", "out.txt", "-s", "1", "x", "-1"],
            ["-l", "-p", "patterns.fasta", "-t", "two.fasta", "-o", "out.txt", "-s", "1", "-1", "-1"],
            ["-l", "-p", "missing.fasta", "-t", "texts.fasta", "-o", "out.txt", "-s", "1", "-1", "-1"],
            ["-l", "-p", "patterns.fasta", "-t", "missing.fasta", "-o", "out.txt", "-s", "1", "-1", "-1"],
            ["-l", "-p", "patterns.fasta", "-t", "texts.fasta", "-o", "nodir/out.txt", "-s", "1", "-1", "-1"],
            ["-l", "-p", "patterns.fasta"],
            ["-g", "-p", "messy_p.fasta", "-t", "messy_t.fasta", "-o", "out.txt", "-s", "1", "-1", "-1"],
            ["-l", "-p", "messy_p.fasta", "-t", "messy_t.fasta", "-o", "out.txt", "-s", "1", "-1", "-1"],
            ["-l", "-s", "1", "-1", "-1", "-o", "out.txt", "-t", "texts.fasta", "-p", "patterns.fasta"],
            ["-g", "-p", "patterns.fasta", "-t", "texts.fasta", "-o", "out.txt", "-s", "1", "-1"],
            ["-g", "-p", "dash.fasta", "-t", "dash.fasta", "-o", "out.txt", "-s", "1", "-1", "-1"],
            ["-l", "-p", "dash.fasta", "-t", "dash.fasta", "-o", "out.txt", "-s", "1", "-1", "-1"],
            ["-g", "-p", "dash_p.fasta", "-t", "dash_t.fasta", "-o", "out.txt", "-s", "1", "-1", "-1"],
            ["-g", "-p", "dash_p.fasta", "-t", "dash_t.fasta", "-o", "out.txt", "-s", "2", "-3", "-5"],
            ["-g", "-p", "big_p.fasta", "-t", "big_t.fasta", "-o", "out.txt", "-s", "100000", "-100000", "-100000"],
            ["-l", "-p", "big_p.fasta", "-t", "big_t.fasta", "-o", "out.txt", "-s", "100000", "-100000", "-100000"],
        ]
        for args in cases:
            outp = os.path.join(td, "out.txt")
            if os.path.exists(outp):
                os.remove(outp)
            pr = subprocess.run([O.REF_CLI] + args, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            cli.append(dict(args=args, rc=pr.returncode, stderr=L(pr.stderr.replace(O.REF_CLI.encode(), b"hw2")),
                            stdout=L(pr.stdout),
                            output=L(open(outp, "rb").read()) if os.path.exists(outp) else None))
    out["cli"] = {"files": files, "cases": cli}

    for key, val in out.items():
        with open(os.path.join(HERE, key + ".json"), "w") as f:
            json.dump(val, f, indent=0 if key in ("random", "edge", "bundled", "dash") else 1)
        print(key, os.path.getsize(os.path.join(HERE, key + ".json")), "bytes")


if __name__ == "__main__":
    main()
