"""hw4 path through the C ABI: NW distance kernel (hw4's tie-break), host UPGMA, hw4-compatible CLI."""
import os
import random
import subprocess

import pytest

import oracle_lib as O
from conftest import B, GOLDEN, load_golden, switched_context

SCORINGS = [(1, -1, -1), (2, -3, -5), (5, -4, -4), (1, -3, -1), (1, 1, 1), (0, 0, 0), (-1, 2, 1), (1, -1, 0), (100, -90, -70)]


def test_upgma_matches_oracle_cpu(pkg):
    """host-only code: runs without a GPU"""
    rng = random.Random(1)
    for n in (1, 2, 3, 5, 9):
        names = [b"s%d" % i for i in range(n)]
        d = [[0.0] * n for _ in range(n)]
        for i in range(n):
            for j in range(i + 1, n):
                d[i][j] = d[j][i] = float(rng.randint(0, 7))   # many ties
        assert pkg.upgma_newick(d, names) == O.upgma(d, names)


@pytest.mark.gpu
def test_distance_reference_fixtures(ctx):
    g = load_golden("hw4")
    by_sc = {}
    for rec in g["pairs"]:
        by_sc.setdefault(tuple(rec["scoring"]), []).append(rec)
    for sc, recs in by_sc.items():
        seqs, pa, pb = [], [], []
        for rec in recs:
            seqs += [B(rec["a"]), B(rec["b"])]
            pa.append(len(seqs) - 2)
            pb.append(len(seqs) - 1)
        assert ctx.distances(seqs, pa, pb, *sc) == [r["dist"] for r in recs], sc
    seqs = [O.gen(1, 2, i, 1000) for i in range(16)]
    pa = [i for i in range(16) for j in range(i + 1, 16)]
    pb = [j for i in range(16) for j in range(i + 1, 16)]
    assert ctx.distances(seqs, pa, pb, 1, -1, -1) == g["gen16x1000"]["dist"]


@pytest.mark.gpu
@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("alphabet", [b"ACGT", b"AC", bytes(range(65, 91)), bytes(range(1, 256))])
def test_distance_random_batches_match_oracle(alphabet, packed):
    """both forms of the kernel: (H, dist) packed into one int32 key (the default while dist < 2^12 and H fits 18 bits;
    the scoring 100/-90/-70 does not and takes the plain form by itself) and the plain two-value form (forced)."""
    rng = random.Random(len(alphabet) + 7)
    lens = [0, 1, 2, 3, 4, 5, 31, 32, 33, 63, 64, 65, 127, 128, 129, 200, 300]
    seqs = [bytes(rng.choice(alphabet) for _ in range(rng.choice(lens) if rng.random() < 0.5 else rng.randint(1, 260)))
            for _ in range(60)]
    pa = [rng.randrange(60) for _ in range(500)]
    pb = [rng.randrange(12) if rng.random() < 0.8 else rng.randrange(60) for _ in range(500)]
    with switched_context(**({} if packed else {"PWA_NO_PACKED_DIST": "1"})) as ctx:
        for sc in SCORINGS:
            got = ctx.distances(seqs, pa, pb, *sc)
            want = [O.nw_distance(seqs[a], seqs[b], *sc)[0] for a, b in zip(pa, pb)]
            bad = [k for k in range(500) if got[k] != want[k]]
            assert not bad, (sc, [(len(seqs[pa[k]]), len(seqs[pb[k]]), got[k], want[k]) for k in bad[:5]])


@pytest.mark.gpu
def test_distance_long_sequences_leave_the_packed_form(ctx):
    """n + m > 4000: distances no longer fit the key's 12 bits; 1000 x 1000 sits inside.  Both against the oracle."""
    seqs = [O.gen(4, 2, i, n) for i, n in enumerate([1000, 1000, 999, 2500, 2400, 1700])]
    pa = [0, 0, 1, 3, 3, 4, 0]
    pb = [1, 2, 2, 4, 5, 5, 3]
    for sc in [(1, -1, -1), (5, -4, -4)]:
        b = ctx.batch_distances(seqs[:3], [0, 0, 1], [1, 2, 2], *sc)
        assert "PACKED" in b.info()["kernel"]
        b.close()
        b = ctx.batch_distances(seqs, pa, pb, *sc)
        assert "PACKED" not in b.info()["kernel"]
        b.close()
        got = ctx.distances(seqs, pa, pb, *sc)
        assert got == [O.nw_distance(seqs[a], seqs[c], *sc)[0] for a, c in zip(pa, pb)], sc


@pytest.mark.gpu
def test_hw4_cli(pkg, tmp_path):
    import test_oracle_hw4
    test_oracle_hw4.run_cli_cases(pkg.CLI4_PATH, tmp_path)
    pr = subprocess.run([pkg.CLI4_PATH, "-i", os.path.join(GOLDEN, "hw4_input.fasta"), "-t", "t.txt", "-s", "1", "-1", "-1"], cwd=tmp_path)
    assert pr.returncode == 0
    assert (tmp_path / "t.txt").read_bytes() == open(os.path.join(GOLDEN, "hw4_tree.txt"), "rb").read()
