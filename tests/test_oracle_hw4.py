"""The hw4 checker (NW distance, UPGMA, CLI) against fixtures generated from the unmodified hw4.cpp.  CPU only."""
import os
import random
import subprocess

import pytest

import oracle_lib as O
from conftest import B, GOLDEN, load_golden


def test_hw4_distance_fixtures():
    g = load_golden("hw4")
    for rec in g["pairs"]:
        assert O.nw_distance(B(rec["a"]), B(rec["b"]), *rec["scoring"])[0] == rec["dist"]
    seqs = [O.gen(1, 2, i, 1000) for i in range(16)]
    tab = [O.nw_distance(seqs[i], seqs[j], 1, -1, -1)[0] for i in range(16) for j in range(i + 1, 16)]
    assert tab == g["gen16x1000"]["dist"]


def run_cli_cases(exe, tmp_path):
    g = load_golden("hw4")
    for case in g["cli"]:
        if "fasta" in case:
            (tmp_path / "in.fa").write_bytes(B(case["fasta"]))
            args = ["-i", "in.fa", "-t", "tree.txt", "-s"] + [str(x) for x in case["scoring"]]
        else:
            args = case["args"]
        out = tmp_path / "tree.txt"
        if out.exists():
            out.unlink()
        pr = subprocess.run([exe] + args, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert pr.returncode == case["rc"], args
        assert pr.stderr.replace(exe.encode(), b"hw4") == B(case["stderr"]), args
        if "tree" in case:
            assert out.read_bytes() == B(case["tree"])


def test_hw4_oracle_cli(tmp_path):
    O.oracle4()
    run_cli_cases(O.ORACLE4_CLI, tmp_path)
    pr = subprocess.run([O.ORACLE4_CLI, "-i", os.path.join(GOLDEN, "hw4_input.fasta"), "-t", "t.txt", "-s", "1", "-1", "-1"], cwd=tmp_path)
    assert pr.returncode == 0
    assert (tmp_path / "t.txt").read_bytes() == open(os.path.join(GOLDEN, "hw4_tree.txt"), "rb").read()


@pytest.mark.skipif(not O.have_ref4(), reason="oracle/_ref only exists in the dev container")
def test_hw4_distance_differential():
    rng = random.Random(44)
    for it in range(1500):
        a = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 50)))
        b = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 50)))
        sc = rng.choice([(1, -1, -1), (2, -3, -5), (1, 1, 1), (0, 0, 0), (-1, 2, 1), (1, -1, 0)])
        assert O.nw_distance(a, b, *sc)[0] == O.ref_nw_distance(a, b, *sc), (a, b, sc)
