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


# ------------------------------------------------------------------ alignments with traceback (hw3.cpp:23-135, 261-283)
def test_affine_alignments_match_reference_vectors(ctx):
    """pwa_align_affine_batch on the 220 pairs whose gapped strings came from the unmodified reference: scores and
    every alignment column (the op list is the two strings, read backwards)."""
    from test_oracle_hw3 import _ops_from_strings
    recs = load_golden("hw3_affine")["alignments"]
    by_sc = {}
    for rec in recs:
        by_sc.setdefault(tuple(rec["scoring"]), []).append(rec)
    for sc, rs in by_sc.items():
        seqs, pa, pb = [], [], []
        for rec in rs:
            seqs += [B(rec["a"]), B(rec["b"])]
            pa.append(len(seqs) - 2)
            pb.append(len(seqs) - 1)
        got = ctx.align_affine_batch(seqs, pa, pb, *sc)
        for g, rec in zip(got, rs):
            assert g["score"] == rec["score"], (sc, rec)
            assert g["ops"] == _ops_from_strings(B(rec["a1"]), B(rec["a2"])), (sc, rec)


@pytest.mark.parametrize("alphabet", [b"ACGT", b"AC", b"ACDEFGHIKLMNPQRSTVWY"])
def test_affine_alignments_against_a_shared_center(ctx, alphabet):
    """the center-star shape: one string1 against many string2 (64 lanes to a wave, several waves), lengths across
    several 32-row strips and not multiples of 4, mutated copies (long diagonal runs with gaps of every kind), empty
    and one-symbol sequences, a second group with another string1 in the same call."""
    rng = random.Random(len(alphabet))
    def mutate(s, rate):
        out = bytearray()
        for c in s:
            r = rng.random()
            if r < rate / 3:
                continue
            if r < 2 * rate / 3:
                out += bytes(rng.choice(alphabet) for _ in range(rng.randint(1, 6)))
            out.append(rng.choice(alphabet) if r > 1 - rate / 3 else c)
        return bytes(out)
    center = bytes(rng.choice(alphabet) for _ in range(211))
    others = [mutate(center, rng.choice([0.01, 0.05, 0.2, 0.6])) for _ in range(90)]
    others += [b"", center[:1], center, center[:33], center[5:70], bytes(rng.choice(alphabet) for _ in range(150))]
    center2 = bytes(rng.choice(alphabet) for _ in range(37))
    seqs = [center, center2] + others
    pa = [0] * len(others) + [1] * 20 + [2]
    pb = list(range(2, 2 + len(others))) + list(range(2, 22)) + [0]
    for sc in [(5, -4, -16, -4), (1, -1, -2, -1), (1, -1, 0, -1), (2, -1, -3, 1), (0, 0, 0, 0)]:
        got = ctx.align_affine_batch(seqs, pa, pb, *sc)
        for k, g in enumerate(got):
            want = O.affine_align(seqs[pa[k]], seqs[pb[k]], *sc)
            assert g["score"] == want["score"], (sc, k, len(seqs[pa[k]]), len(seqs[pb[k]]))
            assert g["ops"] == want["ops"], (sc, k, len(seqs[pa[k]]), len(seqs[pb[k]]))


def test_hw3_cli_matches_reference_cases(pkg, tmp_path):
    """hw3_amd (both dynamic programs on the GPU, FASTA / merge / PHYLIP on the host) against the outputs of the
    unmodified reference program: 35 cases incl. its bundled inputs, odd FASTA files, messages and exit codes."""
    from test_oracle_hw3 import run_hw3_cases
    run_hw3_cases(pkg.HW3_CLI_PATH, tmp_path)


def test_hw3_cli_reproduces_reference_output_phy(pkg, tmp_path):
    import subprocess
    out = tmp_path / "o.phy"
    subprocess.run([pkg.HW3_CLI_PATH, "-i", os.path.join(GOLDEN, "hw3_input.fasta"), "-o", str(out), "-s", "5:-4:-16:-4"], check=True)
    assert out.read_bytes() == open(os.path.join(GOLDEN, "hw3_output.phy"), "rb").read()


def test_hw3_cli_on_16_sequences_of_1000_against_oracle_cli(pkg, tmp_path):
    """the reference's input161000.fasta (16 x 1000 bp): whole pipeline, byte for byte against the oracle's main"""
    import subprocess
    O.oracle3()
    src = os.path.join(GOLDEN, "hw3_input161000.fasta")
    a, b = tmp_path / "a.phy", tmp_path / "b.phy"
    subprocess.run([pkg.HW3_CLI_PATH, "-i", src, "-o", str(a), "-s", "5:-4:-16:-4"], check=True)
    subprocess.run([O.ORACLE3_CLI, "-i", src, "-o", str(b), "-s", "5:-4:-16:-4"], check=True)
    assert a.read_bytes() == b.read_bytes()
