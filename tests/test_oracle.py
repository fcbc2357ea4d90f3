"""The CHECKER checked: oracle/hw2_oracle.c against the reference's golden outputs and the fixtures
generated from the unmodified hw2.cpp (tests/golden/make_golden.py).  CPU only."""
import hashlib
import os
import random
import shutil
import subprocess

import pytest

import oracle_lib as O
from conftest import B, GOLDEN, load_golden


def sha(b):
    return hashlib.sha256(b).hexdigest()


def check_full(rec, got):
    assert got["score"] == rec["score"]
    assert got["cigar"] == B(rec["cigar"])
    assert got["mdz"] == B(rec["mdz"])
    assert got["aligned_pattern"] == B(rec["aligned_pattern"])
    assert got["aligned_reference"] == B(rec["aligned_reference"])
    assert got["overlap"] == rec["overlap"]


@pytest.mark.parametrize("name", ["bundled", "edge", "random", "dash"])
@pytest.mark.parametrize("compact", [False, True])
def test_oracle_matches_reference_fixtures(name, compact):
    for rec in load_golden(name):
        got = O.align(rec["mode"], B(rec["p"]), B(rec["t"]), *rec["scoring"], compact=compact)
        check_full(rec, got)
        s = O.score(rec["mode"], B(rec["p"]), B(rec["t"]), *rec["scoring"])
        assert s[0] == rec["score"] and (s[1], s[2]) == got["end"]


def test_generator_self_check():
    # SURVEY.md 8(d)
    assert O.gen(1, 0, 0, 32) == b"GCAAAATTTCCTCTACCCAATTGGACGCATGC"
    assert O.gen(1, 1, 0, 32) == b"CTTTATTTTGGTCAGTCATTCGTCTGCTTAGT"
    assert O.gen(1, 2, 5, 32) == b"ACAGGAGATCAATATTTTCTTATACACTCACT"


def test_oracle_kats():
    for rec in load_golden("kat"):
        p, t = O.gen(*rec["gen_p"]), O.gen(*rec["gen_t"])
        big = len(p) * len(t) > 2_000_000
        got = O.align(rec["mode"], p, t, *rec["scoring"], compact=big)
        assert got["score"] == rec["score"]
        assert got["overlap"] == rec["overlap"]
        assert len(got["aligned_pattern"]) == rec["aligned_len"]
        assert sha(got["cigar"]) == rec["cigar_sha256"]
        assert sha(got["mdz"]) == rec["mdz_sha256"]
        assert sha(got["aligned_pattern"]) == rec["aligned_pattern_sha256"]
        assert sha(got["aligned_reference"]) == rec["aligned_reference_sha256"]
        assert O.score(rec["mode"], p, t, *rec["scoring"])[0] == rec["score"]


def test_oracle_big_scores_and_real_c4_file():
    """fixtures added in round 2: scores x lengths beyond 2^28 (bigscore.json) and the sibling program's real 16 x 1000 bp
    file (c4_real.json), both from the unmodified hw2.cpp"""
    for rec in load_golden("bigscore"):
        p, t = O.gen(*rec["gen_p"]), O.gen(*rec["gen_t"])
        got = O.align(rec["mode"], p, t, *rec["scoring"], compact=True)
        assert (got["score"], got["overlap"], len(got["aligned_pattern"])) == (rec["score"], rec["overlap"], rec["aligned_len"])
        assert sha(got["cigar"]) == rec["cigar_sha256"] and sha(got["mdz"]) == rec["mdz_sha256"]
    c4r = load_golden("c4_real")
    seqs = O.read_fasta(os.path.join(GOLDEN, c4r["file"]))
    for key, want in c4r["scorings"].items():
        sc = tuple(int(x) for x in key.split(","))
        for mode in ("nw", "sw"):
            assert [O.score(mode, seqs[i], seqs[j], *sc)[0] for i in range(16) for j in range(i + 1, 16)] == want[mode]


def test_survey_known_answers():
    # SURVEY.md 8(a) table, scoring 1/-1/-1, bundled pairs
    pats = [b"GATTACACCCCCCCCCCCCC", b"ATCAAGCGTCGGCATATGGC", b"ATAGC"]
    txts = [b"GTCGACGCATTTTTTTTTTT", b"ATCAGCGATCATCGGCATAT", b"ATATTGC"]
    nw = [(-11, b"1M1D1M1I17M", b"1^A1G2G1A0T0T0T0T0T0T0T0T0T0T0T0", 2), (8, b"3M1D4M4I9M3D", b"3^A13^GGC0", 9),
          (3, b"3M2I2M", b"5", 3)]
    sw = [(3, b"3M", b"3"), (11, b"3M1D4M4I9M", b"3^A13"), (3, b"3M", b"3")]
    for p, t, a, b in zip(pats, txts, nw, sw):
        r = O.align("nw", p, t, 1, -1, -1)
        assert (r["score"], r["cigar"], r["mdz"], r["overlap"]) == a
        r = O.align("sw", p, t, 1, -1, -1)
        assert (r["score"], r["cigar"], r["mdz"]) == b
    # SURVEY.md section 4 edge cases
    r = O.align("sw", b"AAAA", b"CCCC", 1, -1, -1)
    assert (r["score"], r["cigar"], r["mdz"]) == (0, b"", b"0")
    r = O.align("nw", b"AAAA", b"CCCC", 1, -1, -1)
    assert (r["score"], r["cigar"], r["mdz"]) == (-4, b"4M", b"0C0C0C0C0")
    r = O.align("nw", b"AAAA", b"CCCC", 1, -3, -1)
    assert (r["score"], r["cigar"], r["mdz"]) == (-8, b"4D4I", b"0^AAAA0")


def test_batch_tables():
    c3 = load_golden("c3_small")
    pats = [O.gen(1, 0, i, c3["pattern_len"]) for i in range(c3["n_patterns"])]
    txts = [O.gen(1, 1, i, c3["text_len"]) for i in range(c3["n_texts"])]
    for key, tab in c3["scorings"].items():
        sc = tuple(int(x) for x in key.split(","))
        k = 0
        for p in pats[:24]:
            for t in txts:
                assert list(O.score("sw", p, t, *sc)) == tab[k]
                k += 1
            k += 0
    c4 = load_golden("c4_small")
    seqs = [O.gen(1, 2, i, c4["len"]) for i in range(c4["n_seq"])]
    got = [O.score("nw", seqs[i], seqs[j], *c4["scoring"])[0] for i in range(16) for j in range(i + 1, 16)]
    assert got == c4["scores_upper_triangle"] and sum(got) == c4["sum"] == 11397


def _run_cli_cases(exe, tmp_path):
    cli = load_golden("cli")
    for f in ("patterns.fasta", "texts.fasta"):
        shutil.copyfile(os.path.join(GOLDEN, f), tmp_path / f)
    for name, content in cli["files"].items():
        (tmp_path / name).write_bytes(B(content))
    for case in cli["cases"]:
        outp = tmp_path / "out.txt"
        if outp.exists():
            outp.unlink()
        pr = subprocess.run([exe] + case["args"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert pr.returncode == case["rc"], case["args"]
        assert pr.stderr.replace(exe.encode(), b"hw2") == B(case["stderr"]), case["args"]
        assert pr.stdout == B(case["stdout"])
        got = outp.read_bytes() if outp.exists() else None
        want = B(case["output"]) if case["output"] is not None else None
        assert got == want, case["args"]


def test_oracle_cli_matches_reference_cli(tmp_path):
    O.oracle()
    _run_cli_cases(O.ORACLE_CLI, tmp_path)


def test_oracle_cli_reproduces_reference_goldens(tmp_path):
    """The reference's own known-answer files (README.txt:16)."""
    O.oracle()
    for flag, golden in (("-g", "global.txt"), ("-l", "local.txt")):
        out = tmp_path / golden
        rc, err = O.run_cli(O.ORACLE_CLI, [flag, "-p", os.path.join(GOLDEN, "patterns.fasta"), "-t",
                                           os.path.join(GOLDEN, "texts.fasta"), "-o", out, "-s", 1, -1, -1])
        assert rc == 0 and err == b""
        assert out.read_bytes() == open(os.path.join(GOLDEN, golden), "rb").read()


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref (compiled reference) only exists in the dev container")
def test_oracle_differential_against_compiled_reference():
    rng = random.Random(2025)
    scorings = [(1, -1, -1), (2, -3, -5), (5, -4, -4), (1, -3, -1), (0, 0, 0), (1, 1, 1), (-1, 2, 1), (3, -1, 2)]
    for it in range(1500):
        alpha = rng.choice([b"ACGT", b"AC", b"ACGTN"])
        p = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 60)))
        t = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 60)))
        sc = rng.choice(scorings)
        for mode in ("nw", "sw"):
            a, b = O.align(mode, p, t, *sc), O.ref_align(mode, p, t, *sc)
            for k in ("score", "aligned_pattern", "aligned_reference", "cigar", "mdz", "overlap"):
                assert a[k] == b[k], (mode, p, t, sc, k)


def test_oracle_matrices_consistent_with_pinned_alignment():
    """orc_matrices (used for whole-matrix GPU parity) must reproduce the pinned functions: dp[n][m] / max dp
    = score, and walking its traceback matrix by the reference's rules gives the pinned op list."""
    import numpy as np
    rng = random.Random(9)
    for it in range(60):
        p = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 50)))
        t = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 50)))
        sc = rng.choice([(1, -1, -1), (2, -3, -5), (1, 1, 1), (0, 0, 0), (-1, 2, 1)])
        for mode in ("nw", "sw"):
            dp, tb = O.matrices(mode, p, t, *sc)
            a = O.align(mode, p, t, *sc)
            i, j = a["end"]
            assert (dp[len(p), len(t)] if mode == "nw" else dp.max()) == a["score"]
            ops = bytearray()
            while (i > 0 or j > 0) if mode == "nw" else (i > 0 and j > 0 and dp[i, j] != 0):
                c = chr(tb[i, j])
                if c == "d":
                    ops += b"M"; i -= 1; j -= 1
                elif c == "u":
                    ops += b"D"; i -= 1
                else:
                    ops += b"I"; j -= 1
            assert bytes(ops) == a["ops"] and (i, j) == tuple(a["start"])
