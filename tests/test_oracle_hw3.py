"""The hw3 affine checker checked against fixtures generated from the unmodified hw3.cpp.  CPU only."""
import os
import random

import pytest

import oracle_lib as O
from conftest import B, GOLDEN, load_golden


def test_affine_oracle_matches_reference_fixtures():
    g = load_golden("hw3_affine")
    for rec in g["random"]:
        assert O.affine_score(B(rec["a"]), B(rec["b"]), *rec["scoring"]) == rec["score"]
    seqs = [O.gen(1, 2, i, 1000) for i in range(16)]
    tab = [O.affine_score(seqs[i], seqs[j], *g["gen16x1000"]["scoring"]) for i in range(16) for j in range(i + 1, 16)]
    assert tab == g["gen16x1000"]["scores"]
    assert O.affine_score(O.gen(1, 0, 0, 2000), O.gen(1, 1, 0, 3000), *g["gen_2000x3000"]["scoring"]) == g["gen_2000x3000"]["score"]


def test_affine_oracle_on_bundled_inputs():
    g = load_golden("hw3_affine")["bundled"]
    for fname, per in g.items():
        seqs = [s for _, s in O.read_fasta_hw3(os.path.join(GOLDEN, "hw3_" + fname))]
        for key, want in per.items():
            sc = tuple(int(x) for x in key.split(","))
            tab = [O.affine_score(seqs[i], seqs[j], *sc) for i in range(len(seqs)) for j in range(i + 1, len(seqs))]
            assert tab == want["scores"]
            c, sums = O.center(tab, len(seqs))
            assert c == want["center"] and sums == want["star"]


def test_center_of_reference_known_answer():
    """output.phy lists the center first (hw3.cpp:329-331): for input.fasta with 5:-4:-16:-4 it is gi|3211."""
    seqs = O.read_fasta_hw3(os.path.join(GOLDEN, "hw3_input.fasta"))
    tab = [O.affine_score(seqs[i][1], seqs[j][1], 5, -4, -16, -4) for i in range(3) for j in range(i + 1, 3)]
    c, _ = O.center(tab, 3)
    first = open(os.path.join(GOLDEN, "hw3_output.phy"), "rb").read().split(b"\n")[1][:10].strip()
    assert seqs[c][0] == first


@pytest.mark.skipif(not O.have_ref3(), reason="oracle/_ref only exists in the dev container")
def test_affine_oracle_differential():
    rng = random.Random(8)
    for it in range(1500):
        a = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 50)))
        b = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 50)))
        sc = rng.choice([(5, -4, -16, -4), (1, -1, -2, -1), (1, 1, 1, 1), (0, 0, 0, 0), (4, -5, 2, -1), (2, -1, -3, 1)])
        assert O.affine_score(a, b, *sc) == O.ref_affine_score(a, b, *sc), (a, b, sc)


def _ops_from_strings(a1, a2):
    """alignment columns in traceback order from the two gapped strings"""
    return bytes(ord("I") if x == ord("-") else (ord("D") if y == ord("-") else ord("M")) for x, y in zip(a1, a2))[::-1]


def test_affine_alignment_oracle_matches_reference_vectors():
    """hw3.cpp:23-135 with the strings requested: the reference's own gapped strings for 220 pairs (tie-breaks between
    V / F / E and between opening and extending a gap all matter here)."""
    for rec in load_golden("hw3_affine")["alignments"]:
        got = O.affine_align(B(rec["a"]), B(rec["b"]), *rec["scoring"])
        assert (got["score"], got["a1"], got["a2"]) == (rec["score"], B(rec["a1"]), B(rec["a2"])), rec
        assert got["ops"] == _ops_from_strings(B(rec["a1"]), B(rec["a2"]))


@pytest.mark.skipif(not O.have_ref3(), reason="oracle/_ref only exists in the dev container")
def test_affine_alignment_oracle_differential():
    rng = random.Random(18)
    for it in range(600):
        alpha = rng.choice([b"AC", b"ACGT", b"ACDEFGHIKLMNPQRSTVWY"])
        a = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 45)))
        b = bytes(rng.choice(alpha) for _ in range(rng.randint(0, 45)))
        sc = rng.choice([(5, -4, -16, -4), (1, -1, -2, -1), (1, -1, 0, -1), (0, 0, 0, 0), (2, -1, -3, 1), (1, 1, 1, 1)])
        r, o = O.ref_affine_align(a, b, *sc), O.affine_align(a, b, *sc)
        assert (o["score"], o["a1"], o["a2"]) == (r["score"], r["a1"], r["a2"]), (a, b, sc)


def run_hw3_cases(exe, tmp_path, name="hw3"):
    """every case of tests/golden/hw3_cli.json (outputs of the unmodified reference program) through `exe`"""
    import shutil
    import subprocess
    cli = load_golden("hw3_cli")
    for f in ("hw3_input.fasta", "hw3_input16100.fasta", "hw3_input41000.fasta"):
        shutil.copyfile(os.path.join(GOLDEN, f), tmp_path / f)
    for fname, content in cli["files"].items():
        (tmp_path / fname).write_bytes(B(content))
    for case in cli["cases"]:
        outp = tmp_path / "out.phy"
        if outp.exists():
            outp.unlink()
        pr = subprocess.run([exe] + case["args"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert pr.returncode == case["rc"], (case["args"], pr.stderr)
        assert pr.stdout.replace(exe.encode(), b"hw3") == B(case["stdout"]), case["args"]
        got = outp.read_bytes() if outp.exists() else None
        want = B(case["output"]) if case["output"] is not None else None
        assert got == want, case["args"]


def test_hw3_oracle_cli_matches_reference_cases(tmp_path):
    """the oracle's restatement of hw3's main (FASTA quirks, center, gap-pattern merge, PHYLIP writer, messages)"""
    O.oracle3()   # builds oracle/ if needed
    run_hw3_cases(O.ORACLE3_CLI, tmp_path)


def test_hw3_oracle_cli_reproduces_reference_output_phy(tmp_path):
    import subprocess
    O.oracle3()
    out = tmp_path / "o.phy"
    subprocess.run([O.ORACLE3_CLI, "-i", os.path.join(GOLDEN, "hw3_input.fasta"), "-o", str(out), "-s", "5:-4:-16:-4"], check=True)
    assert out.read_bytes() == open(os.path.join(GOLDEN, "hw3_output.phy"), "rb").read()
