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
