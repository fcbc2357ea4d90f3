"""Parity of the affine (hw3) score kernel through the C ABI against the oracle and the fixtures generated
from the unmodified hw3.cpp.  Needs an MI355X."""
import os
import random

import pytest

import oracle_lib as O
from conftest import B, GOLDEN, load_golden

pytestmark = pytest.mark.gpu

SCORINGS = [(5, -4, -16, -4), (1, -1, -2, -1), (2, -3, -5, -2), (1, -1, 0, -1), (3, -2, -1, -3), (1, 1, 1, 1), (0, 0, 0, 0),
            (4, -5, 2, -1), (100, -90, -300, -70)]


def test_affine_reference_fixtures(ctx):
    g = load_golden("hw3_affine")
    by_sc = {}
    for rec in g["random"]:
        by_sc.setdefault(tuple(rec["scoring"]), []).append(rec)
    for sc, recs in by_sc.items():
        seqs, pa, pb = [], [], []
        for rec in recs:
            seqs += [B(rec["a"]), B(rec["b"])]
            pa.append(len(seqs) - 2)
            pb.append(len(seqs) - 1)
        got = ctx.scores_affine(seqs, pa, pb, *sc)
        assert got == [r["score"] for r in recs], sc


def test_affine_bundled_all_pairs_and_center(ctx):
    g = load_golden("hw3_affine")["bundled"]
    for fname, per in g.items():
        seqs = [s for _, s in O.read_fasta_hw3(os.path.join(GOLDEN, "hw3_" + fname))]
        pa = [i for i in range(len(seqs)) for j in range(i + 1, len(seqs))]
        pb = [j for i in range(len(seqs)) for j in range(i + 1, len(seqs))]
        for key, want in per.items():
            sc = tuple(int(x) for x in key.split(","))
            got = ctx.scores_affine(seqs, pa, pb, *sc)
            assert got == want["scores"], (fname, sc)
            c, sums = O.center(got, len(seqs))
            assert c == want["center"] and sums == want["star"]


def test_affine_generator_kats(ctx):
    g = load_golden("hw3_affine")
    seqs = [O.gen(1, 2, i, 1000) for i in range(16)]
    pa = [i for i in range(16) for j in range(i + 1, 16)]
    pb = [j for i in range(16) for j in range(i + 1, 16)]
    got = ctx.scores_affine(seqs, pa, pb, *g["gen16x1000"]["scoring"])
    assert got == g["gen16x1000"]["scores"]
    got = ctx.scores_affine([O.gen(1, 0, 0, 2000), O.gen(1, 1, 0, 3000)], [0], [1], *g["gen_2000x3000"]["scoring"])
    assert got == [g["gen_2000x3000"]["score"]]


@pytest.mark.parametrize("alphabet", [b"ACGT", bytes(range(65, 91)), bytes(range(1, 256))])
def test_affine_random_batches_match_oracle(ctx, alphabet):
    rng = random.Random(len(alphabet) + 1)
    lens = [0, 1, 2, 3, 4, 5, 31, 32, 33, 51, 52, 53, 104, 105, 200, 300]
    seqs = [bytes(rng.choice(alphabet) for _ in range(rng.choice(lens) if rng.random() < 0.5 else rng.randint(1, 260)))
            for _ in range(60)]
    pa = [rng.randrange(60) for _ in range(500)]
    pb = [rng.randrange(12) if rng.random() < 0.8 else rng.randrange(60) for _ in range(500)]
    for sc in SCORINGS:
        got = ctx.scores_affine(seqs, pa, pb, *sc)
        want = [O.affine_score(seqs[a], seqs[b], *sc) for a, b in zip(pa, pb)]
        bad = [k for k in range(500) if got[k] != want[k]]
        assert not bad, (sc, [(len(seqs[pa[k]]), len(seqs[pb[k]]), got[k], want[k]) for k in bad[:5]])


def test_affine_many_strips(ctx):
    pats = [O.gen(3, 0, i, n) for i, n in enumerate([1000, 999, 520, 105, 1])]
    txts = [O.gen(3, 1, i, m) for i, m in enumerate([1500, 1501, 1502, 1503, 3])]
    seqs = pats + txts
    pa = [i for i in range(len(pats)) for _ in txts]
    pb = [len(pats) + j for _ in pats for j in range(len(txts))]
    for sc in [(5, -4, -16, -4), (1, -1, -2, -1), (100, -90, -300, -70)]:
        got = ctx.scores_affine(seqs, pa, pb, *sc)
        assert got == [O.affine_score(seqs[a], seqs[b], *sc) for a, b in zip(pa, pb)], sc
