"""Parity of the HIP path (through the C ABI) against the oracle and the committed reference fixtures.
Every test here needs a real MI355X: run with `pytest -m gpu`."""
import hashlib
import os
import random
import shutil
import subprocess

import pytest

import oracle_lib as O
from conftest import B, GOLDEN, load_golden, switched_context

pytestmark = pytest.mark.gpu

SCORINGS = [(1, -1, -1), (2, -3, -5), (5, -4, -4), (1, -3, -1), (0, 0, 0), (1, 1, 1), (-1, 2, 1), (3, -1, 2),
            (1, -1, 0), (2, -1, -3)]


def sha(b):
    return hashlib.sha256(b).hexdigest()


def check_alignment(got, want):
    """got: Context.align dict; want: oracle dict or fixture record."""
    for k in ("score", "cigar", "mdz", "aligned_pattern", "aligned_reference", "overlap"):
        w = want[k]
        if isinstance(w, str):
            w = B(w)
        assert got[k] == w, (k, got[k], w)


# ------------------------------------------------------------------ full alignments (fill + traceback)
@pytest.mark.parametrize("name", ["bundled", "edge", "random", "dash"])
def test_align_matches_reference_fixtures(ctx, name):
    recs = load_golden(name)
    for rec in recs:
        got = ctx.align(rec["mode"], B(rec["p"]), B(rec["t"]), *rec["scoring"])
        check_alignment(got, rec)


def test_align_ops_and_cells_match_oracle(ctx):
    rng = random.Random(11)
    for it in range(120):
        alpha = rng.choice([b"ACGT", b"AC", bytes(range(65, 91))])
        p = bytes(rng.choice(alpha) for _ in range(rng.randint(1, 700)))
        t = bytes(rng.choice(alpha) for _ in range(rng.randint(1, 700)))
        sc = rng.choice(SCORINGS)
        for mode in ("nw", "sw"):
            got = ctx.align(mode, p, t, *sc, raw=True)
            want = O.align(mode, p, t, *sc)
            assert got["score"] == want["score"]
            assert got["ops"] == want["ops"]
            assert tuple(got["end"]) == tuple(want["end"])
            assert tuple(got["start"]) == tuple(want["start"])


def test_align_kats_long_pairs(ctx):
    """SURVEY.md 8(d) generator KATs incl. C2 (10k x 10k): strings compared by sha256 of the
    reference's output."""
    for rec in load_golden("kat"):
        p, t = O.gen(*rec["gen_p"]), O.gen(*rec["gen_t"])
        got = ctx.align(rec["mode"], p, t, *rec["scoring"])
        assert got["score"] == rec["score"]
        assert got["overlap"] == rec["overlap"]
        assert len(got["aligned_pattern"]) == rec["aligned_len"]
        assert sha(got["cigar"]) == rec["cigar_sha256"]
        assert sha(got["mdz"]) == rec["mdz_sha256"]
        assert sha(got["aligned_pattern"]) == rec["aligned_pattern_sha256"]
        assert sha(got["aligned_reference"]) == rec["aligned_reference_sha256"]


@pytest.mark.parametrize("alphabet,table", [(b"ACGT", True), (b"ACGT", False), (b"ACGTNRY", True), (b"ACDEFGHIKLMNPQRSTVWY", True)])
def test_align_batch_matches_oracle(alphabet, table):
    """both scoring forms of the traceback fill: byte-table lookups on coded sequences (alphabets of <= 7 symbols) and
    compare + select on raw bytes (larger alphabets, or forced)."""
    rng = random.Random(5)
    seqs = [bytes(rng.choice(alphabet) for _ in range(rng.randint(0, 400))) for _ in range(40)]
    pa = [rng.randrange(40) for _ in range(100)]
    pb = [rng.randrange(40) for _ in range(100)]
    with switched_context(**({} if table else {"PWA_NO_PAIR_TABLE": "1"})) as c:
        for mode in ("nw", "sw"):
            for sc in [(1, -1, -1), (2, -3, -5), (20, -15, -9), (40, -3, 2)]:   # the last two leave the byte table's range
                res = c.align_batch(mode, seqs, pa, pb, *sc)
                for k, r in enumerate(res):
                    want = O.align(mode, seqs[pa[k]], seqs[pb[k]], *sc)
                    assert r["score"] == want["score"], (mode, k)
                    assert r["ops"] == want["ops"], (mode, k)
                    assert tuple(r["end"]) == tuple(want["end"])
                    assert tuple(r["start"]) == tuple(want["start"])


def test_align_batch_with_score_band_matches_oracle(ctx):
    """pwa_ctx_set_score_band: the fill also writes the int32 score band to HBM (5 B/cell, the reference's own
    footprint, hw2.cpp:193-194); alignments must not change -- several chunks' worth of pairs, both modes."""
    rng = random.Random(8)
    seqs = [bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 300))) for _ in range(30)]
    seqs += [O.gen(8, 1, 0, 3000), O.gen(8, 0, 0, 700)]
    pa = [rng.randrange(32) for _ in range(80)] + [31]
    pb = [rng.randrange(32) for _ in range(80)] + [30]
    ctx.set_score_band(True)
    try:
        for mode in ("nw", "sw"):
            res = ctx.align_batch(mode, seqs, pa, pb, 2, -3, -5)
            for k, r in enumerate(res):
                want = O.align(mode, seqs[pa[k]], seqs[pb[k]], 2, -3, -5)
                assert (r["score"], r["ops"], tuple(r["end"]), tuple(r["start"])) == \
                    (want["score"], want["ops"], tuple(want["end"]), tuple(want["start"])), (mode, k)
    finally:
        ctx.set_score_band(False)


def _mutate(rng, s, rate):
    out = bytearray()
    for c in s:
        r = rng.random()
        if r < rate / 3:
            continue                                   # deletion
        if r < 2 * rate / 3:
            out.append(rng.choice(b"ACGT"))            # insertion
        out.append(rng.choice(b"ACGT") if r > 1 - rate / 3 else c)
    return bytes(out)


def test_device_overlaps_match_oracle(ctx, pkg):
    """pwa_overlaps: overlapLongestExactMatch (hw2.cpp:267-278) computed by the device walk, with no op list, for
    every pair -- random pairs (short runs), mutated copies (runs much longer than one 64-lane trip, runs crossing
    traceback windows and stripes), identical sequences, empty sides; global and local walks."""
    rng = random.Random(99)
    seqs = [bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 300))) for _ in range(24)]
    base = [bytes(rng.choice(b"ACGT") for _ in range(n)) for n in (64, 65, 129, 500, 1000, 2500)]
    seqs += base
    seqs += [_mutate(rng, b, rate) for b in base for rate in (0.002, 0.02, 0.2)]
    seqs += [b"", b"A", b"ACGT" * 40, b"ACGT" * 40 + b"T"]
    n = len(seqs)
    pa = [rng.randrange(n) for _ in range(150)] + [24 + i for i in range(6) for _ in range(4)] + [n - 2, n - 2, n - 4]
    pb = [rng.randrange(n) for _ in range(150)] + [k for i in range(6) for k in (24 + i, 30 + 3 * i, 31 + 3 * i, 32 + 3 * i)] \
        + [n - 2, n - 1, n - 3]
    for mode in ("nw", "sw"):
        for sc in [(1, -1, -1), (2, -3, -5)]:
            scores, ovl = ctx.overlaps(mode, seqs, pa, pb, *sc)
            for k in range(len(pa)):
                want = O.align(mode, seqs[pa[k]], seqs[pb[k]], *sc)
                assert scores[k] == want["score"], (mode, sc, k)
                assert ovl[k] == want["overlap"], (mode, sc, k, len(seqs[pa[k]]), len(seqs[pb[k]]), ovl[k], want["overlap"])
    # the same walk with op lists agrees with itself
    res = ctx.align_batch("nw", seqs, pa, pb, 1, -1, -1)
    scores, ovl = ctx.overlaps("nw", seqs, pa, pb, 1, -1, -1)
    for k, r in enumerate(res):
        assert r["score"] == scores[k]
        assert pkg.alignment_overlap(seqs[pa[k]], seqs[pb[k]], r["ops"], r["end"]) == ovl[k]


def test_device_overlaps_with_dash_symbols_match_reference(ctx):
    """A literal '-' inside a SEQUENCE: overlapLongestExactMatch (hw2.cpp:269) requires both symbols of a column to be
    non-'-', so such a column ends a run although it is a diagonal move over equal symbols.  Fixture = outputs of the
    unmodified reference (tests/golden/dash.json), through pwa_overlaps in ONE batch per mode and scoring -- both arena
    forms: coded symbols (alphabets of <= 7) and, with the table switched off, raw bytes."""
    recs = load_golden("dash")
    groups = {}
    for rec in recs:
        groups.setdefault((rec["mode"], tuple(rec["scoring"])), []).append(rec)
    for raw in (False, True):
        with switched_context(**({"PWA_NO_PAIR_TABLE": "1"} if raw else {})) as c:
            for (mode, sc), rs in groups.items():
                seqs = [B(r["p"]) for r in rs] + [B(r["t"]) for r in rs]
                pa = list(range(len(rs)))
                pb = [len(rs) + k for k in range(len(rs))]
                scores, ovl = c.overlaps(mode, seqs, pa, pb, *sc)
                for k, r in enumerate(rs):
                    assert scores[k] == r["score"], (mode, sc, k)
                    assert ovl[k] == r["overlap"], (mode, sc, r["p"], r["t"], ovl[k], r["overlap"])
    # the case of VERDICT r01: the winner of -g changes (pair 2, overlap 4; pair 1's run is cut at the '-': 2)
    scores, ovl = ctx.overlaps("nw", [b"AC-GT", b"ACGA"], [0, 1], [0, 1], 1, -1, -1)
    assert (scores, ovl) == ([5, 4], [2, 4])


# ------------------------------------------------------------------ scores-only batches
def test_scores_c3_small_fixture(sctx):
    ctx = sctx
    c3 = load_golden("c3_small")
    pats = [O.gen(1, 0, i, c3["pattern_len"]) for i in range(c3["n_patterns"])]
    txts = [O.gen(1, 1, i, c3["text_len"]) for i in range(c3["n_texts"])]
    seqs = pats + txts
    pa, pb = [], []
    for i in range(len(pats)):
        for j in range(len(txts)):
            pa.append(i)
            pb.append(len(pats) + j)
    for key, tab in c3["scorings"].items():
        sc = tuple(int(x) for x in key.split(","))
        got = ctx.scores("sw", seqs, pa, pb, *sc)
        assert got == [r[0] for r in tab]
        s, ei, ej = ctx.scores("sw", seqs, pa, pb, *sc, want_end=True)
        assert [list(x) for x in zip(s, ei, ej)] == tab


def test_scores_c4_small_fixture(sctx):
    ctx = sctx
    c4 = load_golden("c4_small")
    seqs = [O.gen(1, 2, i, c4["len"]) for i in range(c4["n_seq"])]
    pa, pb = [], []
    for i in range(16):
        for j in range(i + 1, 16):
            pa.append(i)
            pb.append(j)
    got = ctx.scores("nw", seqs, pa, pb, *c4["scoring"])
    assert got == c4["scores_upper_triangle"] and sum(got) == 11397
    # the plain-H kernel (gap-shift not applicable with these magnitudes) must agree as well
    got2 = ctx.scores("nw", seqs, pa, pb, 100, -90, -70)
    want2 = [O.score("nw", seqs[a], seqs[b], 100, -90, -70)[0] for a, b in zip(pa, pb)]
    assert got2 == want2


@pytest.mark.parametrize("alphabet", [b"ACGT", b"ACGTN", bytes(range(65, 91)), bytes(range(1, 256))])
def test_scores_random_batches_match_oracle(sctx, alphabet):
    ctx = sctx
    rng = random.Random(len(alphabet))
    lens = [0, 1, 2, 3, 4, 5, 63, 64, 65, 127, 128, 129, 151, 152, 153, 200, 300, 520]
    seqs = []
    for _ in range(70):
        n = rng.choice(lens) if rng.random() < 0.5 else rng.randint(1, 330)
        seqs.append(bytes(rng.choice(alphabet) for _ in range(n)))
    seqs.append(seqs[3])
    pa = [rng.randrange(len(seqs)) for _ in range(700)]
    pb = [rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]) if rng.random() < 0.8 else rng.randrange(len(seqs))
          for _ in range(700)]
    for sc in SCORINGS:
        for mode in ("nw", "sw"):
            got = ctx.scores(mode, seqs, pa, pb, *sc)
            want = [O.score(mode, seqs[a], seqs[b], *sc)[0] for a, b in zip(pa, pb)]
            bad = [k for k in range(len(pa)) if got[k] != want[k]]
            assert not bad, (mode, sc, bad[:5], [(len(seqs[pa[k]]), len(seqs[pb[k]]), got[k], want[k]) for k in bad[:5]])


def test_scores_end_cells_match_oracle(ctx):
    rng = random.Random(77)
    seqs = [bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 600))) for _ in range(30)]
    pa = [rng.randrange(30) for _ in range(150)]
    pb = [rng.randrange(30) for _ in range(150)]
    for sc in [(1, -1, -1), (1, 1, 1), (0, 0, 0), (2, -3, -5)]:
        for mode in ("nw", "sw"):
            s, ei, ej = ctx.scores(mode, seqs, pa, pb, *sc, want_end=True)
            for k in range(len(pa)):
                w = O.score(mode, seqs[pa[k]], seqs[pb[k]], *sc)
                assert (s[k], ei[k], ej[k]) == tuple(w), (mode, sc, k)


@pytest.mark.parametrize("engine", ["auto", "stripes"])
def test_scores_end_cells_with_ties_on_both_engines(engine):
    """End cells of a scores pass: short patterns over a coded alphabet take the band-less mini-stripe kernels, everything else (and
    everything under PWA_TB_ENGINE=0) the stripe engine.  Two-letter sequences put the maximum in many cells: the first one in
    row-major order is the answer (hw2.cpp:225-229); a pattern-only symbol ('N') equals no text symbol on either engine."""
    rng = random.Random(4242)
    pats = [bytes(rng.choice(b"AC") for _ in range(n)) for n in (1, 2, 15, 16, 17, 63, 64, 65, 100, 150, 160, 161, 255, 256, 257, 300)]
    pats += [bytes(rng.choice(b"ACN") for _ in range(150)), b"A" * 150, b"AC" * 75]
    txts = [bytes(rng.choice(b"AC") for _ in range(m)) for m in (1, 5, 15, 16, 17, 31, 32, 33, 200, 1000, 1003)] + [b"A" * 333, b"CA" * 170]
    seqs = pats + txts
    pa = [i for i in range(len(pats)) for _ in txts]
    pb = [len(pats) + j for _ in pats for j in range(len(txts))]
    with switched_context(**({"PWA_TB_ENGINE": "0"} if engine == "stripes" else {})) as c:
        for sc in [(1, -1, -1), (2, -3, -5), (1, 0, 0), (3, 1, -2)]:
            for mode in ("nw", "sw"):
                b = c.batch(mode, seqs, pa, pb, *sc, want_end=True)
                try:
                    assert ("mini_fill_kernel" in b.info()["kernel"]) == (engine == "auto"), b.info()["kernel"]
                    b.run()
                    s, ei, ej = b.fetch()
                finally:
                    b.close()
                for k in range(len(pa)):
                    w = O.score(mode, seqs[pa[k]], seqs[pb[k]], *sc)
                    assert (s[k], ei[k], ej[k]) == tuple(w), (mode, sc, k, len(seqs[pa[k]]), len(seqs[pb[k]]))


def test_scores_end_cells_with_scores_beyond_the_byte_table(ctx):
    """End cells over a DNA alphabet (coded arena) with scores whose key constants do not fit the byte table: no mini-stripe kernels, the
    stripe engine's plain compare form runs on the codes (a pattern-only symbol is code 7: equal to no text code)."""
    rng = random.Random(682)
    seqs = [bytes(rng.choice(b"ACGT") for _ in range(rng.choice([1, 16, 100, 150, 257, 600]))) for _ in range(24)] + [b"ACGTNNNNAC" * 15]
    pa = [rng.randrange(25) for _ in range(120)]
    pb = [rng.randrange(24) for _ in range(120)]   # ('N' never in a text)
    for mode in ("nw", "sw"):
        b = ctx.batch(mode, seqs, pa, pb, 100, -90, -70, want_end=True)
        try:
            assert "mini_fill_kernel" not in b.info()["kernel"] and "keyed" not in b.info()["kernel"], b.info()["kernel"]
            b.run()
            s, ei, ej = b.fetch()
        finally:
            b.close()
        for k in range(len(pa)):
            assert (s[k], ei[k], ej[k]) == tuple(O.score(mode, seqs[pa[k]], seqs[pb[k]], 100, -90, -70)), (mode, k)


def test_scores_end_cells_of_a_chip_filling_list_of_long_patterns(ctx):
    """Enough multi-stripe pairs to fill the chip: the stripe engine takes them at RL = 4 (fewer, taller stripes) instead of the
    RL = 2 it gives a handful of long pairs; same scores and end cells (oracle)."""
    rng = random.Random(99)
    alphabet = bytes(range(65, 91))
    pats = [bytes(rng.choice(alphabet[:3]) for _ in range(rng.randint(1900, 2100))) for _ in range(300)]
    txts = [bytes(rng.choice(alphabet[:3]) for _ in range(rng.randint(250, 330))) for _ in range(300)]
    txts[0] = alphabet * 12   # > 7 text symbols: no coded arena, no mini-stripe kernels
    seqs = pats + txts
    pa = list(range(300))
    pb = [300 + i for i in range(300)]
    for mode in ("sw", "nw"):
        b = ctx.batch(mode, seqs, pa, pb, 2, -3, -2, want_end=True)
        try:
            assert "pair_fill_kernel<RL=4" in b.info()["kernel"], b.info()["kernel"]
            b.run()
            s, ei, ej = b.fetch()
        finally:
            b.close()
        for k in range(300):
            assert (s[k], ei[k], ej[k]) == tuple(O.score(mode, seqs[pa[k]], seqs[pb[k]], 2, -3, -2)), (mode, k)


def test_scores_many_strips_and_long_texts(sctx):
    ctx = sctx
    """patterns spanning several register strips, texts with every length residue mod 4."""
    pats = [O.gen(3, 0, i, n) for i, n in enumerate([1000, 999, 700, 513, 512, 511, 129, 1])]
    txts = [O.gen(3, 1, i, m) for i, m in enumerate([2000, 2001, 2002, 2003, 5])]
    seqs = pats + txts
    pa = [i for i in range(len(pats)) for _ in txts]
    pb = [len(pats) + j for _ in pats for j in range(len(txts))]
    for sc in [(1, -1, -1), (5, -4, -4), (100, -90, -70)]:
        for mode in ("nw", "sw"):
            got = ctx.scores(mode, seqs, pa, pb, *sc)
            want = [O.score(mode, seqs[a], seqs[b], *sc)[0] for a, b in zip(pa, pb)]
            assert got == want, (mode, sc)


def test_two_strip_tasks_through_the_lds_paired_kernel(ctx):
    """opt-in form (PWA_PAIRED=1): local-alignment batches whose tasks have one or two strips run as two waves per task
    with the strip boundary row in an LDS ring (batch_scores_pair_kernel): every text length residue mod 4, texts shorter than
    one block, longer than the ring, a few single-strip tasks, ragged lanes, many more tasks than workgroups."""
    rng = random.Random(4242)
    txt_lens = [1, 2, 3, 4, 5, 6, 7, 61, 62, 63, 64, 65, 66, 67, 68, 69, 255, 256, 257, 1000, 1001, 1002, 1003]
    txts = [O.gen(11, 1, i, m) for i, m in enumerate(txt_lens)]
    for lo, hi, scoring, n_short in [(140, 152, (1, -1, -1), 2), (95, 104, (2, -3, -5), 1), (125, 150, (5, -4, -4), 0)]:
        pats = [O.gen(11, 0, i, rng.randint(lo, hi)) for i in range(90)]
        pats += [O.gen(11, 2, i, rng.randint(1, lo // 2)) for i in range(6)]
        seqs = pats + txts
        pa, pb = [], []
        for j in range(len(txts)):
            short_text = j < n_short   # a task made only of short patterns is a single strip
            for i in range(len(pats)):
                if (i >= 90) == short_text and rng.random() < 0.9:
                    pa.append(i)
                    pb.append(len(pats) + j)
        with switched_context(PWA_PAIRED="1", PWA_SCORES_ROUTE="0") as c:   # opt-in form (a switch of the context)
            b = c.batch("sw", seqs, pa, pb, *scoring)
            assert "pair_kernel" in b.info()["kernel"], b.info()
            for _ in range(2):
                b.run()
                got = b.fetch()
            b.close()
        want = [O.score("sw", seqs[a], seqs[c], *scoring)[0] for a, c in zip(pa, pb)]
        bad = [k for k in range(len(pa)) if got[k] != want[k]]
        assert not bad, (scoring, bad[:5], [(len(seqs[pa[k]]), len(seqs[pb[k]]), got[k], want[k]) for k in bad[:5]])


@pytest.mark.parametrize("alphabet,expect_lanes", [(b"ACGT", True), (b"ACGTN", True), (b"ACGTNXY", False),
                                                   (bytes(range(65, 91)), True), (bytes(range(0, 256)), False)])
def test_index_paired_lists_use_per_lane_texts(sctx, alphabet, expect_lanes):
    """the reference's own shape -- pattern i against reference i (hw2.cpp:328-338), every pair its own text: local
    scores come from the LANES kernels (each lane streams its own text; columns past a lane's text and rows past its
    pattern are padded with two different never-matching symbols).  Ragged lengths in both directions, texts of every
    length residue, 1..3 strips, empty and one-symbol sequences, an alphabet with no free code and one with no free
    byte (those fall back to the other engines and must still be exact)."""
    rng = random.Random(len(alphabet) * 7 + 1)
    n_pairs = 700
    pats, txts = [], []
    for k in range(n_pairs):
        n = rng.choice([0, 1, 2, 75, 76, 77, 150, 152, 153, 200]) if rng.random() < 0.3 else rng.randint(1, 230)
        m = rng.choice([0, 1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257]) if rng.random() < 0.3 else rng.randint(1, 700)
        pats.append(bytes(rng.choice(alphabet) for _ in range(n)))
        txts.append(bytes(rng.choice(alphabet) for _ in range(m)))
    # make a few pairs share long common substrings so that maxima sit deep inside the matrices
    for k in range(0, n_pairs, 9):
        if len(txts[k]) > 40:
            cut = rng.randrange(len(txts[k]) - 30)
            pats[k] = pats[k][:10] + txts[k][cut:cut + 30] + pats[k][10:]
    seqs = pats + txts
    pa = list(range(n_pairs))
    pb = [n_pairs + k for k in range(n_pairs)]
    for sc in [(1, -1, -1), (2, -3, -5), (5, -4, -4), (3, 0, 0)]:
        b = sctx.batch("sw", seqs, pa, pb, *sc)
        kern = b.info()["kernel"]
        b.run()
        got = b.fetch()
        b.close()
        if sctx.route == "strips":
            assert ("LANES" in kern) == expect_lanes, (kern, sc)
        want = [O.score("sw", seqs[a], seqs[c], *sc)[0] for a, c in zip(pa, pb)]
        bad = [k for k in range(n_pairs) if got[k] != want[k]]
        assert not bad, (kern, sc, bad[:5], [(len(pats[k]), len(txts[k]), got[k], want[k]) for k in bad[:5]])


@pytest.mark.parametrize("alphabet,pattern_extra,expect_lanes", [(b"ACGT", b"", True), (b"AC", b"", True), (b"ACGTN", b"", False),
                                                                 (b"ACGT", b"N", False)])
def test_index_paired_global_scores_use_right_aligned_lanes(sctx, alphabet, pattern_extra, expect_lanes):
    """global alignment scores of index-paired lists: LANES kernels with right-aligned texts whose front padding the
    table scores like a gap column (alphabets of <= 4 symbols that contain every pattern symbol); other inputs keep
    the text-grouped form.  Ragged lengths both ways, every length residue mod 4, 1..3 strips, one-symbol and empty
    sequences; scorings with gap 0 and with tables that no longer fit a byte (those leave the gap-shifted form)."""
    rng = random.Random(len(alphabet) * 11 + len(pattern_extra))
    n_pairs = 600
    pats, txts = [], []
    for k in range(n_pairs):
        n = rng.choice([0, 1, 2, 63, 64, 65, 128, 152, 153, 300]) if rng.random() < 0.3 else rng.randint(1, 330)
        m = rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 255, 256, 257]) if rng.random() < 0.3 else rng.randint(1, 600)
        pats.append(bytes(rng.choice(alphabet + pattern_extra) for _ in range(n)))
        txts.append(bytes(rng.choice(alphabet) for _ in range(m)))
    for k in range(0, n_pairs, 7):   # related pairs: long diagonals
        if len(txts[k]) > 30:
            pats[k] = txts[k][3:] if k % 2 else txts[k][:len(txts[k]) // 2] + pats[k][:20]
    seqs = pats + txts
    pa = list(range(n_pairs))
    pb = [n_pairs + k for k in range(n_pairs)]
    for sc, shifted in [((1, -1, -1), True), ((2, -3, -5), True), ((5, -4, -4), True), ((1, -1, 0), True), ((3, 0, -2), True),
                        ((100, -90, -70), False)]:
        b = sctx.batch("nw", seqs, pa, pb, *sc)
        kern = b.info()["kernel"]
        b.run()
        got = b.fetch()
        b.close()
        if sctx.route == "strips":
            assert ("LANES" in kern) == (expect_lanes and shifted), (kern, sc)
        want = [O.score("nw", seqs[a], seqs[c], *sc)[0] for a, c in zip(pa, pb)]
        bad = [k for k in range(n_pairs) if got[k] != want[k]]
        assert not bad, (kern, sc, bad[:5], [(len(pats[k]), len(txts[k]), got[k], want[k]) for k in bad[:5]])


def test_index_paired_short_patterns_single_strip_lanes(sctx):
    ctx = sctx
    """per-lane texts with every pattern inside one register strip (the hand-off free LANES instantiation)"""
    rng = random.Random(31)
    n_pairs = 500
    pats = [bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 40))) for _ in range(n_pairs)]
    txts = [bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 900))) for _ in range(n_pairs)]
    seqs = pats + txts
    pa = list(range(n_pairs))
    pb = [n_pairs + k for k in range(n_pairs)]
    for sc in [(1, -1, -1), (2, -3, -5)]:
        b = ctx.batch("sw", seqs, pa, pb, *sc)
        assert "LANES" in b.info()["kernel"] or sctx.route != "strips", b.info()
        b.run()
        got = b.fetch()
        b.close()
        assert got == [O.score("sw", seqs[a], seqs[c], *sc)[0] for a, c in zip(pa, pb)], sc


def test_batch_object_reuse(sctx, pkg):
    ctx = sctx
    seqs = [O.gen(9, 0, i, 150) for i in range(130)] + [O.gen(9, 1, 0, 777)]
    pa = list(range(130))
    pb = [130] * 130
    b = ctx.batch("sw", seqs, pa, pb, 1, -1, -1)
    info = b.info()
    assert info["cells"] == 130 * 150 * 777 and info["padded_cells"] >= info["cells"]
    b.run()
    first = b.fetch()
    b.run()
    assert b.fetch() == first
    assert b.last_ms() > 0
    assert first == [O.score("sw", s, seqs[130], 1, -1, -1)[0] for s in seqs[:130]]
    b.close()


def test_scores_pass_is_routed_by_work(ctx):
    """r03: the scores pass over hw2.cpp's pair loop (328-338) picks its engine from the WORK, pair by pair.  A list of few
    long pairs leaves the lane-per-pair strip kernels (one 10k x 10k pair was one lane of one wave: 779 ms against 1.7 ms),
    a list of many short pairs plus a few long ones is split, and results land in caller order whatever the split --
    scores against the oracle, both modes, and the same lists forced onto either engine."""
    rng = random.Random(303)
    longs = [(O.gen(30, 0, i, 1000 + 100 * i), O.gen(30, 1, i, 1200 - 100 * i)) for i in range(3)]   # ("long": 7+ register strips on one lane)
    shorts = [(O.gen(31, 0, i, rng.randint(20, 150)), O.gen(31, 1, i, rng.randint(100, 400))) for i in range(30000)]

    def lists():
        yield "3 long only", longs, True, False
        mixed = shorts[:400] + [longs[0]] + shorts[400:] + [longs[2]]
        yield "30000 short + 2 long", mixed, True, None    # the long ones on stripes; the short ones wherever the model puts them (strips or mini)
        yield "1 long", longs[:1], True, False
        yield "1000 short", shorts[:1000], None, None      # (16 wave tasks: cheaper off the strips; whatever the model says, exact)

    for name, pairs, want_stripes, want_strips in lists():
        seqs = [p for p, _ in pairs] + [t for _, t in pairs]
        pa = list(range(len(pairs)))
        pb = [len(pairs) + k for k in range(len(pairs))]
        for mode in ("nw", "sw"):
            want = [O.score(mode, p, t, 1, -1, -1)[0] for p, t in pairs]
            b = ctx.batch(mode, seqs, pa, pb, 1, -1, -1)
            kern = b.info()["kernel"]
            b.run()
            got = b.fetch()
            b.close()
            assert got == want, (name, mode, kern)
            if want_stripes is not None:
                assert ("pair_fill_kernel" in kern) == want_stripes, (name, mode, kern)
            if want_strips is not None:
                assert ("batch_scores_kernel" in kern) == want_strips, (name, mode, kern)
            if name.startswith("30000"):
                assert "batch_scores_kernel" in kern or "mini_fill_kernel" in kern, (name, mode, kern)
            for route in ("0", "1"):
                with switched_context(PWA_SCORES_ROUTE=route) as c:
                    assert c.scores(mode, seqs, pa, pb, 1, -1, -1) == want, (name, mode, route)
    # short patterns that leave the strips run on the mini-stripe engine without a band (four pairs per wave), one launch per row class;
    # scorings whose key constants leave the byte table, and alphabets of more than 7 symbols, stay on the stripe engine -- all exact
    few = [(O.gen(32, 0, i, rng.choice([5, 64, 65, 100, 150, 160, 161, 250, 256])), O.gen(32, 1, i, rng.randint(300, 900))) for i in range(1500)]
    few.append((O.gen(32, 0, 9999, 300), O.gen(32, 1, 9999, 700)))   # 300 rows: no mini class
    seqs = [p for p, _ in few] + [t for _, t in few]
    pa = list(range(len(few)))
    pb = [len(few) + k for k in range(len(few))]
    for mode in ("nw", "sw"):
        for sc, want_mini in (((1, -1, -1), True), ((2, -3, -5), True), ((3, 0, -2), True), ((1, -1, 0), True), ((20, -15, -9), True), ((40, -3, 2), False)):   # the last one: (40 - 2) * 4 + 2 leaves the byte table
            b = ctx.batch(mode, seqs, pa, pb, *sc)
            kern = b.info()["kernel"]
            b.run()
            got = b.fetch()
            b.close()
            assert ("mini_fill_kernel" in kern) == want_mini, (mode, sc, kern)
            if want_mini:
                assert kern.count("mini_fill_kernel") >= 4 and "pair_fill_kernel" in kern, kern   # several row classes + the 300-row pair
            want = [O.score(mode, p, t, *sc)[0] for p, t in few]
            bad = [k for k in range(len(few)) if got[k] != want[k]]
            assert not bad, (mode, sc, kern, [(len(few[k][0]), len(few[k][1]), got[k], want[k]) for k in bad[:5]])
    # the device score vector of a split batch is complete in caller-owned memory too (what bench.py hands to the all-gather)
    import torch
    pairs = shorts[:20000] + [longs[1]]
    seqs = [p for p, _ in pairs] + [t for _, t in pairs] + [b""]
    pa = list(range(len(pairs))) + [0]
    pb = [len(pairs) + k for k in range(len(pairs))] + [2 * len(pairs)]
    b = ctx.batch("nw", seqs, pa, pb, 2, -3, -5)
    assert b.info()["kernel"].count(" + ") >= 1, b.info()   # a split batch: several engines write into the one score vector
    t = torch.full((len(pa),), -777, dtype=torch.int32, device="cuda")
    b.set_d_scores(t.data_ptr())
    b.run()
    b.last_ms()
    assert t.cpu().tolist() == [O.score("nw", seqs[a], seqs[c], 2, -3, -5)[0] for a, c in zip(pa, pb)]
    b.close()


# ------------------------------------------------------------------ size-independent properties at larger sizes
def _path_score(p, t, ops, end, match, mismatch, gap):
    i, j = end
    tot = 0
    for o in ops:
        if o == ord("M"):
            i -= 1
            j -= 1
            tot += match if p[i] == t[j] else mismatch
        elif o == ord("D"):
            i -= 1
            tot += gap
        else:
            j -= 1
            tot += gap
    return tot, (i, j)


def test_large_pair_properties(ctx):
    """20k x 20k: the returned path must re-score to the returned score, which must equal the
    oracle's score-only DP; NW == transposed NW; SW path starts where the walk says."""
    p, t = O.gen(1, 0, 7, 20000), O.gen(1, 1, 7, 20000)
    for mode in ("nw", "sw"):
        got = ctx.align(mode, p, t, 1, -1, -1, raw=True)
        want = O.score(mode, p, t, 1, -1, -1)
        assert got["score"] == want[0]
        assert tuple(got["end"]) == (want[1], want[2])
        tot, start = _path_score(p, t, got["ops"], got["end"], 1, -1, -1)
        assert tot == got["score"]
        assert start == tuple(got["start"])
    assert ctx.align("nw", t, p, 1, -1, -1, raw=True)["score"] == ctx.align("nw", p, t, 1, -1, -1, raw=True)["score"]


# ------------------------------------------------------------------ the hw2-compatible CLI
def test_cli_reproduces_reference_goldens(pkg, tmp_path):
    for flag, golden in (("-g", "global.txt"), ("-l", "local.txt")):
        out = tmp_path / golden
        rc, err = O.run_cli(pkg.CLI_PATH, [flag, "-p", os.path.join(GOLDEN, "patterns.fasta"), "-t",
                                           os.path.join(GOLDEN, "texts.fasta"), "-o", out, "-s", 1, -1, -1])
        assert rc == 0 and err == b"", err
        assert out.read_bytes() == open(os.path.join(GOLDEN, golden), "rb").read()


def test_cli_cases_match_reference(pkg, tmp_path):
    cli = load_golden("cli")
    for f in ("patterns.fasta", "texts.fasta"):
        shutil.copyfile(os.path.join(GOLDEN, f), tmp_path / f)
    for name, content in cli["files"].items():
        (tmp_path / name).write_bytes(B(content))
    exe = pkg.CLI_PATH
    for case in cli["cases"]:
        outp = tmp_path / "out.txt"
        if outp.exists():
            outp.unlink()
        pr = subprocess.run([exe] + case["args"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert pr.returncode == case["rc"], (case["args"], pr.stderr)
        assert pr.stderr.replace(exe.encode(), b"hw2") == B(case["stderr"]), case["args"]
        got = outp.read_bytes() if outp.exists() else None
        want = B(case["output"]) if case["output"] is not None else None
        assert got == want, case["args"]


def test_dropin_functions_return_the_reference_fields(pkg):
    """INTEGRATION.md Option B compiled (host/hw2_dropin.cpp -> libhw2_dropin.so): globalAlignmentNeedlemanWunsch /
    localAlignmentSmithWaterman with the reference's exact signatures (hw2.cpp:118, 192) returning a caller-owned
    AlignmentResult* (17-23).  host/hw2_dropin_check calls them as the reference's loop does; all five fields of every
    bundled / edge / dash fixture record (outputs of the unmodified hw2.cpp) must come back."""
    exe = os.path.join(os.path.dirname(pkg.CLI_PATH), "hw2_dropin_check")
    recs = [r for name in ("bundled", "edge", "dash") for r in load_golden(name)]
    inp = b"".join(b"%s %d %d %d %d %d\n" % (b"g" if r["mode"] == "nw" else b"l", *r["scoring"], len(B(r["p"])), len(B(r["t"])))
                   + B(r["p"]) + B(r["t"]) + b"\n" for r in recs)
    pr = subprocess.run([exe], input=inp, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert pr.returncode == 0, pr.stderr
    out, pos = pr.stdout, 0
    for r in recs:
        eol = out.index(b"\n", pos)
        score, la, lc, lm = (int(x) for x in out[pos:eol].split())
        pos = eol + 1
        ap, ar = out[pos:pos + la], out[pos + la:pos + 2 * la]
        cg, md = out[pos + 2 * la:pos + 2 * la + lc], out[pos + 2 * la + lc:pos + 2 * la + lc + lm]
        pos += 2 * la + lc + lm + 1
        assert (score, ap, ar, cg, md) == (r["score"], B(r["aligned_pattern"]), B(r["aligned_reference"]), B(r["cigar"]), B(r["mdz"])), r
    assert pos == len(out)


def test_cli_many_pairs_against_oracle_cli(pkg, tmp_path):
    rng = random.Random(3)
    with open(tmp_path / "p.fa", "w") as fp, open(tmp_path / "t.fa", "w") as ft:
        for i in range(60):
            p = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 200)))
            t = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 400)))
            fp.write(">p%d\n%s\n" % (i, p))
            ft.write(">t%d\n%s\n" % (i, t))
    O.oracle()
    for flag in ("-g", "-l"):
        for sc in [(1, -1, -1), (2, -3, -5)]:
            a, b = tmp_path / "a.txt", tmp_path / "b.txt"
            args = [flag, "-p", "p.fa", "-t", "t.fa", "-s", *sc]
            rc1, _ = O.run_cli(pkg.CLI_PATH, args + ["-o", a], cwd=tmp_path)
            rc2, _ = O.run_cli(O.ORACLE_CLI, args + ["-o", b], cwd=tmp_path)
            assert rc1 == rc2 == 0
            assert a.read_bytes() == b.read_bytes()


def test_cli_sharded_over_devices_matches_oracle_cli(pkg, tmp_path):
    """--devices a,b,c: contiguous blocks of the pair list, one host thread + context per listed GPU (here the same GPU
    several times: the threading, the block arithmetic incl. more devices than pairs, and the first-best selection
    over the merged vectors); ties between blocks must still go to the first pair."""
    rng = random.Random(13)
    with open(tmp_path / "p.fa", "w") as fp, open(tmp_path / "t.fa", "w") as ft:
        for i in range(203):
            p = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 160)))
            t = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 300)))
            if i in (40, 150):   # the same best pair twice, in different blocks: the first one must win
                p, t = "ACGTACGTTTGACCAGTACCAGTT" * 3, "GG" + "ACGTACGTTTGACCAGTACCAGTT" * 3 + "CA"
            fp.write(">p%d\n%s\n" % (i, p))
            ft.write(">t%d\n%s\n" % (i, t))
    with open(tmp_path / "p2.fa", "w") as fp, open(tmp_path / "t2.fa", "w") as ft:
        fp.write(">a\nACGTAC\n>b\nTTGACA\n")
        ft.write(">a\nACGAC\n>b\nTTGCA\n")
    O.oracle()
    for flag in ("-g", "-l"):
        b = tmp_path / "b.txt"
        args = [flag, "-p", "p.fa", "-t", "t.fa", "-s", 1, -1, -1]
        assert O.run_cli(O.ORACLE_CLI, args + ["-o", b], cwd=tmp_path)[0] == 0
        for devs in ("0,0", "0,0,0,0,0", "0"):
            a = tmp_path / "a.txt"
            rc, err = O.run_cli(pkg.CLI_PATH, args + ["-o", a, "--devices", devs], cwd=tmp_path)
            assert rc == 0, err
            assert a.read_bytes() == b.read_bytes(), (flag, devs)
        args2 = [flag, "-p", "p2.fa", "-t", "t2.fa", "-s", 1, -1, -1]
        assert O.run_cli(O.ORACLE_CLI, args2 + ["-o", b], cwd=tmp_path)[0] == 0
        a = tmp_path / "a.txt"
        rc, err = O.run_cli(pkg.CLI_PATH, args2 + ["-o", a, "--devices", "0,0,0,0"], cwd=tmp_path)   # more devices than pairs
        assert rc == 0, err
        assert a.read_bytes() == b.read_bytes()
    rc, err = O.run_cli(pkg.CLI_PATH, ["-l", "-p", "p.fa", "-t", "t.fa", "-s", 1, -1, -1, "-o", tmp_path / "a.txt", "--devices", "0,99"],
                        cwd=tmp_path)
    assert rc == 2 and b"no usable gfx950 device" in err or b"failed" in err


def test_batches_driven_from_other_threads(pkg):
    """every batch entry point sets its context's device itself, so a batch may be run / fetched on a thread other
    than the one that created it, and two contexts may be driven from two threads at once (one GPU here: both on device 0)."""
    import threading
    rng = random.Random(21)
    seqs = [bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 200))) for _ in range(30)]
    pa = [rng.randrange(30) for _ in range(200)]
    pb = [rng.randrange(30) for _ in range(200)]
    want = [O.score("sw", seqs[a], seqs[b], 1, -1, -1)[0] for a, b in zip(pa, pb)]
    c1, c2 = pkg.Context(0), pkg.Context(0)
    b1 = c1.batch("sw", seqs, pa, pb, 1, -1, -1)
    out = {}

    def other(name, batch):
        batch.run()
        out[name] = (batch.fetch(), batch.last_ms() > 0)

    def whole(name, c):
        out[name] = c.scores("sw", seqs, pa, pb, 1, -1, -1)

    th = [threading.Thread(target=other, args=("t1", b1)), threading.Thread(target=whole, args=("t2", c2))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert out["t1"] == (want, True) and out["t2"] == want
    b1.close()
    c1.close()
    c2.close()


def test_device_score_vector_in_caller_memory(sctx):
    ctx = sctx
    """pwa_batch_set_d_scores: kernels write into a torch tensor (what bench.py hands to the RCCL all-gather);
    both engines, including pairs with an empty side."""
    import torch
    seqs = [O.gen(5, 0, i, 100 + i) for i in range(40)] + [b"", O.gen(5, 1, 0, 500)]
    pa = list(range(41))
    pb = [41] * 41
    for mode, sc in (("sw", (1, -1, -1)), ("nw", (1, -1, -1)), ("sw", (1, 1, 1))):   # (1,1,1) -> pair engine
        b = ctx.batch(mode, seqs, pa, pb, *sc)
        t = torch.full((len(pa),), -12345, dtype=torch.int32, device="cuda")
        b.set_d_scores(t.data_ptr())
        b.run()
        b.last_ms()
        want = [O.score(mode, seqs[a], seqs[bb], *sc)[0] for a, bb in zip(pa, pb)]
        assert t.cpu().tolist() == want, (mode, sc)
        assert b.fetch() == want
        b.close()


@pytest.mark.parametrize("engine", ["mini", "stripes", "wide"])
def test_whole_matrices_match_reference_matrices(engine):
    """Every cell of the int32 score band and of the traceback band == the reference's `dp` and `traceback`
    matrices (hw2.cpp:119-156 / 193-231), incl. row/column 0, several stripes, all scorings.  "mini": patterns of up to 256 rows
    through the mini-stripe engine (every RL class, lengths at both ends of each), longer ones through the stripe engine;
    "stripes" (PWA_TB_ENGINE=0): everything through the stripe engine."""
    import numpy as np
    rng = random.Random(42)
    cases = [(1, 1), (3, 70), (70, 3), (255, 300), (256, 64), (257, 65), (600, 777), (1100, 90)]
    if engine == "mini":
        cases += [(64, 100), (65, 31), (96, 200), (97, 16), (128, 129), (129, 15), (150, 333), (160, 47), (161, 48), (192, 17), (193, 250), (16, 1), (17, 500)]
    if engine == "wide":   # PWA_TB_ENGINE=2: one pair per wave for 257 .. 1024 rows (RL = 8 | 16), as batches of such patterns run
        cases = [(257, 65), (300, 700), (384, 50), (385, 49), (511, 64), (512, 100), (513, 40), (768, 20), (769, 21), (1023, 63), (1024, 33),
                 (1000, 1100), (1025, 70)]   # RL = 6 / 8 / 12 / 16 classes at both ends; 1025: stripe engine
    with switched_context(**({"PWA_TB_ENGINE": "0"} if engine == "stripes" else {"PWA_TB_ENGINE": "2"} if engine == "wide" else {})) as c:
        for (n, m) in cases:
            p = bytes(rng.choice(b"ACGT") for _ in range(n))
            t = bytes(rng.choice(b"ACGT") for _ in range(m))
            for sc in rng.sample(SCORINGS, 4):
                for mode in ("nw", "sw"):
                    dp, tb = c.matrices(mode, p, t, *sc)
                    wdp, wtb = O.matrices(mode, p, t, *sc)
                    assert np.array_equal(dp, wdp), (mode, n, m, sc)
                    assert np.array_equal(tb, wtb), (mode, n, m, sc)


def test_scores_degenerate_shapes(sctx):
    ctx = sctx
    """empty batch, texts shorter than one 4-column block, very long pattern against tiny texts,
    duplicate pairs, a pattern aligned with itself."""
    assert ctx.scores("sw", [b"ACGT"], [], [], 1, -1, -1) == []
    long_p = O.gen(11, 0, 0, 5000)
    seqs = [long_p, b"A", b"AC", b"ACG", b"ACGT", b"ACGTA", b"", O.gen(11, 1, 0, 301)]
    pa = [0, 0, 0, 0, 0, 0, 0, 7, 7, 0, 1, 2, 3, 4, 5, 7, 7]
    pb = [1, 2, 3, 4, 5, 6, 7, 7, 7, 0, 1, 1, 1, 1, 1, 1, 6]
    for sc in [(1, -1, -1), (2, -3, -5), (1, 1, 1)]:
        for mode in ("nw", "sw"):
            got = ctx.scores(mode, seqs, pa, pb, *sc)
            want = [O.score(mode, seqs[a], seqs[b], *sc)[0] for a, b in zip(pa, pb)]
            assert got == want, (mode, sc, got, want)
            s, ei, ej = ctx.scores(mode, seqs, pa, pb, *sc, want_end=True)
            assert [(x, y, z) for x, y, z in zip(s, ei, ej)] == [tuple(O.score(mode, seqs[a], seqs[b], *sc)) for a, b in zip(pa, pb)]


def test_align_degenerate_shapes(ctx):
    for p, t in [(b"A", b"A"), (b"A", b"C"), (b"", b"ACGT"), (b"ACGT", b""), (b"", b""), (O.gen(2, 0, 0, 1300), b"ACG"),
                 (b"ACG", O.gen(2, 1, 0, 1300)), (O.gen(2, 0, 1, 257), O.gen(2, 0, 1, 257))]:
        for sc in [(1, -1, -1), (1, -3, -1), (1, 1, 1)]:
            for mode in ("nw", "sw"):
                got = ctx.align(mode, p, t, *sc)
                want = O.align(mode, p, t, *sc)
                check_alignment(got, want)
                assert got["ops"] == want["ops"] and tuple(got["end"]) == tuple(want["end"])


def test_error_reporting(ctx, pkg):
    with pytest.raises(pkg.PwaError):
        ctx.scores("sw", [b"ACGT"], [0], [5], 1, -1, -1)        # pair index out of range
    import ctypes as C
    L = pkg.lib()
    score = C.c_int32()
    n_ops = C.c_uint64()
    ops = C.create_string_buffer(4)
    rc = L.pwa_align(ctx._h, 1, 1, -1, -1, b"ACGTACGT", 8, b"ACGTACGT", 8, C.byref(score), ops, 4, C.byref(n_ops), None, None)
    assert rc == -5 and b"ops_cap" in L.pwa_last_error(ctx._h)   # PWA_E_CAPACITY
    rc = L.pwa_align(ctx._h, 7, 1, -1, -1, b"A", 1, b"A", 1, C.byref(score), ops, 4, C.byref(n_ops), None, None)
    assert rc == -1


# ------------------------------------------------------------------ BASELINE.json full sizes, size-independent properties
def test_full_size_c3_batch_properties(ctx):
    """C3 at full size (4096 x 150 bp  X  256 x 10 kbp = 1 048 576 pairs): a seeded sample of 384 pairs equals the
    oracle, every score respects 0 <= s <= 150, and the result does not depend on the order of the pair list."""
    import numpy as np
    pats = [O.gen(1, 0, p, 150) for p in range(4096)]
    txts = [O.gen(1, 1, t, 10000) for t in range(256)]
    seqs = pats + txts
    pa = np.repeat(np.arange(4096, dtype=np.uint32), 256)
    pb = np.tile(np.arange(256, dtype=np.uint32) + np.uint32(4096), 4096)
    b = ctx.batch("sw", seqs, pa, pb, 1, -1, -1)
    assert b.info()["kernel"] == "batch_scores_kernel<R=76,BM_SWS,SC_PERM>", b.info()   # the work-aware routing keeps a full list on the strips
    b.run()
    s = b.fetch(numpy_out=True)
    b.close()
    assert s.min() >= 0 and s.max() <= 150
    rng = np.random.default_rng(3)
    for k in rng.choice(len(pa), 384, replace=False):
        assert int(s[k]) == O.score("sw", seqs[pa[k]], seqs[pb[k]], 1, -1, -1)[0], k
    perm = rng.permutation(len(pa))
    b2 = ctx.batch("sw", seqs, np.ascontiguousarray(pa[perm]), np.ascontiguousarray(pb[perm]), 1, -1, -1)
    b2.run()
    s2 = b2.fetch(numpy_out=True)
    b2.close()
    assert np.array_equal(s2, s[perm])
    assert int(s.astype(np.int64).sum()) == 35376135   # checksum of the bench line (bit-exact vs hw2.cpp on its sample)


def test_full_size_c5_pair_properties(ctx):
    """C5 at full size (NW 100k x 100k, 10 GB traceback band): the returned path is a complete global path
    (#M+#D = n, #M+#I = m, ends in (0,0)), it re-scores to the returned score, NW(p,t) == NW(t,p), and the
    score equals the oracle's score-only DP."""
    n = m = 100000
    p, t = O.gen(1, 0, 0, n), O.gen(1, 1, 0, m)
    got = ctx.align("nw", p, t, 1, -1, -1, raw=True)
    ops = got["ops"]
    assert tuple(got["end"]) == (n, m) and tuple(got["start"]) == (0, 0)
    assert ops.count(b"M") + ops.count(b"D") == n and ops.count(b"M") + ops.count(b"I") == m
    tot, start = _path_score(p, t, ops, got["end"], 1, -1, -1)
    assert tot == got["score"] and start == (0, 0)
    assert ctx.align("nw", t, p, 1, -1, -1, raw=True)["score"] == got["score"]
    assert got["score"] == O.score("nw", p, t, 1, -1, -1)[0] == 11391   # the oracle's O(m)-memory DP: ~15 s of CPU


def test_full_size_c5_matches_the_reference_run(ctx):
    """C5 string for string: score, overlap, aligned length and the sha256 of CIGAR, MD:Z and both gapped strings that the
    UNMODIFIED hw2.cpp produced for gen(1,0,0,100000) x gen(1,1,0,100000), 1/-1/-1 -- one 50 GB, ten-minute run of
    globalAlignmentNeedlemanWunsch and one of localAlignmentSmithWaterman in the build container
    (tests/golden/make_golden_c5.py -> kat_c5.json).  A tie-break slip that preserves the score would show here."""
    for rec in load_golden("kat_c5"):
        p, t = O.gen(*rec["gen_p"]), O.gen(*rec["gen_t"])
        got = ctx.align(rec["mode"], p, t, *rec["scoring"])
        assert got["score"] == rec["score"]
        assert got["overlap"] == rec["overlap"]
        assert len(got["aligned_pattern"]) == rec["aligned_len"]
        assert (len(got["cigar"]), sha(got["cigar"])) == (rec["cigar_len"], rec["cigar_sha256"])
        assert (len(got["mdz"]), sha(got["mdz"])) == (rec["mdz_len"], rec["mdz_sha256"])
        assert sha(got["aligned_pattern"]) == rec["aligned_pattern_sha256"]
        assert sha(got["aligned_reference"]) == rec["aligned_reference_sha256"]


def test_full_size_c4_all_pairs_properties(ctx):
    """C4 at full size (all 523 776 pairs of 1024 generator sequences x 1000 bp, NW 1/-1/-1): a seeded sample of 512 pairs
    equals the oracle, NW(a,b) == NW(b,a) for every pair (the transposed list is a different schedule: other texts, other
    lanes), the first 120 pairs are the committed 16-sequence table (sum 11 397), scores stay in range."""
    import numpy as np
    n_seq = 1024
    seqs = [O.gen(1, 2, i, 1000) for i in range(n_seq)]
    ii, jj = np.triu_indices(n_seq, k=1)
    pa, pb = ii.astype(np.uint32), jj.astype(np.uint32)
    assert len(pa) == 523776
    b = ctx.batch("nw", seqs, pa, pb, 1, -1, -1)
    assert b.info()["kernel"].startswith("batch_scores_kernel<R=128,BM_NWG,SC_PERM>"), b.info()   # routed by work: the strips keep (nearly) all of it
    b.run()
    s = b.fetch(numpy_out=True)
    b.close()
    assert s.min() >= -1000 and s.max() <= 1000
    rng = np.random.default_rng(4)
    for k in rng.choice(len(pa), 512, replace=False):
        assert int(s[k]) == O.score("nw", seqs[pa[k]], seqs[pb[k]], 1, -1, -1)[0], k
    b2 = ctx.batch("nw", seqs, pb, pa, 1, -1, -1)
    b2.run()
    s2 = b2.fetch(numpy_out=True)
    b2.close()
    assert np.array_equal(s, s2)
    c4 = load_golden("c4_small")
    sub = [int(s[k]) for k in range(len(pa)) if pa[k] < 16 and pb[k] < 16]
    assert sub == c4["scores_upper_triangle"] and sum(sub) == 11397
    print("c4 checksum", int(s.astype(np.int64).sum()))


def test_c4_real_sequences_match_reference(ctx, pkg):
    """all pairs of the sibling program's real 16 x ~1000 bp file (tests/golden/hw3_input161000.fasta =
    Multiple_Sequence_Alignment/input161000.fasta), NW and SW, 1/-1/-1 and 5/-4/-4, per-pair against the unmodified
    hw2.cpp (c4_real.json) and the survey's sums (NW 3013 / 82 730, min -600 / -2150, max 400 / 2400; SW 11 231 / 102 928)."""
    c4r = load_golden("c4_real")
    blob, off, first = pkg.read_fasta(os.path.join(GOLDEN, c4r["file"]))
    seqs = [blob[off[k]:off[k + 1]] for k in range(len(off) - 1)]
    assert len(seqs) == 16
    pa = [i for i in range(16) for j in range(i + 1, 16)]
    pb = [j for i in range(16) for j in range(i + 1, 16)]
    for key, want in c4r["scorings"].items():
        sc = tuple(int(x) for x in key.split(","))
        nw = ctx.scores("nw", seqs, pa, pb, *sc)
        sw = ctx.scores("sw", seqs, pa, pb, *sc)
        assert nw == want["nw"] and sw == want["sw"]
        assert (sum(nw), min(nw), max(nw), sum(sw)) == (want["nw_sum"], want["nw_min"], want["nw_max"], want["sw_sum"])
    assert c4r["scorings"]["1,-1,-1"]["nw_sum"] == 3013 and c4r["scorings"]["5,-4,-4"]["nw_sum"] == 82730
    assert c4r["scorings"]["1,-1,-1"]["sw_sum"] == 11231 and c4r["scorings"]["5,-4,-4"]["sw_sum"] == 102928


def test_scores_beyond_the_packed_key_range_match_reference(ctx):
    """scores x lengths > 2^28: the traceback fill leaves its packed H*4+priority keys for the plain int32 form
    (pair_fill_kernel<..., KEYED = false>) instead of refusing -- the reference's recurrences hold for anything its int
    holds (hw2.cpp:140-153, 206-222).  Fixture: bigscore.json from the unmodified hw2.cpp."""
    for rec in load_golden("bigscore"):
        p, t = O.gen(*rec["gen_p"]), O.gen(*rec["gen_t"])
        got = ctx.align(rec["mode"], p, t, *rec["scoring"])
        assert got["score"] == rec["score"], rec["scoring"]
        assert got["overlap"] == rec["overlap"]
        assert len(got["aligned_pattern"]) == rec["aligned_len"]
        assert sha(got["cigar"]) == rec["cigar_sha256"] and sha(got["mdz"]) == rec["mdz_sha256"]
        assert sha(got["aligned_pattern"]) == rec["aligned_pattern_sha256"]
        assert sha(got["aligned_reference"]) == rec["aligned_reference_sha256"]
    # whole matrices in that form, cell by cell, and the -g selection inputs of a small batch
    p, t = O.gen(9, 0, 0, 700), O.gen(9, 1, 0, 900)
    for mode in ("nw", "sw"):
        dp, tb = ctx.matrices(mode, p, t, 300000, -250000, -200000)
        wdp, wtb = O.matrices(mode, p, t, 300000, -250000, -200000)
        assert (dp == wdp).all() and (tb == wtb).all()
    seqs = [O.gen(9, 0, i, 1400 + 37 * i) for i in range(6)] + [O.gen(9, 1, i, 1500 - 29 * i) for i in range(6)]
    pa, pb = list(range(6)), list(range(6, 12))
    scores, ovl = ctx.overlaps("nw", seqs, pa, pb, 100000, -100000, -100000)
    for k in range(6):
        want = O.align("nw", seqs[pa[k]], seqs[pb[k]], 100000, -100000, -100000)
        assert (scores[k], ovl[k]) == (want["score"], want["overlap"])


@pytest.mark.parametrize("name", ["bundled", "edge", "dash"])
def test_plain_int32_traceback_form_on_the_reference_fixtures(name):
    """the same fixtures as test_align_matches_reference_fixtures through the plain int32 traceback form (forced)"""
    with switched_context(PWA_NO_KEYED_TB="1") as c:
        for rec in load_golden(name):
            got = c.align(rec["mode"], B(rec["p"]), B(rec["t"]), *rec["scoring"])
            check_alignment(got, rec)


def test_one_shot_calls_cut_oversized_pair_lists_into_arena_chunks(ctx):
    """pwa_scores / pwa_distances / pwa_scores_affine on pair lists whose sequences exceed one 4 GiB arena: the list is
    processed in runs of pairs that fit (limit lowered to 6 KiB here so that ~30 chunks form, incl. sequences reused
    across chunks and a pair of one sequence with itself)."""
    rng = random.Random(77)
    seqs = [bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 900))) for _ in range(60)]
    pa = [rng.randrange(60) for _ in range(300)] + [5]
    pb = [rng.randrange(60) for _ in range(300)] + [5]
    whole = {m: ctx.scores(m, seqs, pa, pb, 2, -3, -5) for m in ("nw", "sw")}
    whole_end = ctx.scores("sw", seqs, pa, pb, 2, -3, -5, want_end=True)
    whole_d = ctx.distances(seqs, pa, pb, 1, -1, -1)
    whole_a = ctx.scores_affine_oneshot(seqs, pa, pb, 5, -4, -16, -4)
    with switched_context(PWA_ARENA_LIMIT="6144") as c:
        for m in ("nw", "sw"):
            got = c.scores(m, seqs, pa, pb, 2, -3, -5)
            assert got == whole[m] == [O.score(m, seqs[a], seqs[b], 2, -3, -5)[0] for a, b in zip(pa, pb)]
        assert c.scores("sw", seqs, pa, pb, 2, -3, -5, want_end=True) == whole_end
        assert c.distances_oneshot(seqs, pa, pb, 1, -1, -1) == whole_d == [O.nw_distance(seqs[a], seqs[b], 1, -1, -1)[0] for a, b in zip(pa, pb)]
        assert c.scores_affine_oneshot(seqs, pa, pb, 5, -4, -16, -4) == whole_a == [O.affine_score(seqs[a], seqs[b], 5, -4, -16, -4) for a, b in zip(pa, pb)]


def test_one_shot_calls_halve_runs_that_exceed_a_batch_object_limit():
    """the arena estimate is not the only 32-bit limit inside a batch object -- the right-aligned per-lane text rows of the
    global LANES form have their own (here lowered to 64 KiB through PWA_LANE_ROWS_LIMIT).  pwa_batch_create refuses such a list
    (PWA_E_CAPACITY); the one-shot call halves the run until it fits and must still return every score."""
    rng = random.Random(962)
    n = 900
    pats = [bytes(rng.choice(b"ACGT") for _ in range(rng.randint(20, 120))) for _ in range(n)]
    txts = [bytes(rng.choice(b"ACGT") for _ in range(rng.randint(100, 500))) for _ in range(n)]
    seqs = pats + txts
    pa, pb = list(range(n)), [n + k for k in range(n)]
    want = [O.score("nw", seqs[a], seqs[b], 1, -1, -1)[0] for a, b in zip(pa, pb)]
    with switched_context(PWA_LANE_ROWS_LIMIT="65536", PWA_SCORES_ROUTE="0") as c:
        with pytest.raises(Exception, match="capacity"):
            c.batch("nw", seqs, pa, pb, 1, -1, -1)
        assert c.scores("nw", seqs, pa, pb, 1, -1, -1) == want


def test_one_shot_scores_are_pipelined_over_runs(ctx):
    """r03: the runs of a one-shot score call are prepared (coded, uploaded) while the previous run computes.  A ~100 MB
    list cut into six runs (PWA_PIPE_RUNS=6; by default only lists beyond one 4 GiB arena are cut) must give the results of the single
    run and of the strictly serial form (PWA_NO_PIPELINE) pair for pair, and the oracle's on a sample.  Index-paired (the reference's
    own loop, hw2.cpp:328-338) plus pairs that share a text."""
    import numpy as np
    import bench
    n = 9000
    pats = [bench.gen(50, 0, i, 100 + i % 60) for i in range(n)]
    txts = [bench.gen(50, 1, i, 11000 + 37 * (i % 50)) for i in range(n)]   # ~100 MB of texts
    seqs = pats + txts
    pa = list(range(n)) + [5, 6, 7]
    pb = [n + k for k in range(n)] + [n + 1, n + 1, n + 1]
    for mode in ("sw", "nw"):
        got = ctx.scores(mode, seqs, pa, pb, 1, -1, -1)
        with switched_context(PWA_PIPE_RUNS="6") as c:
            assert c.scores(mode, seqs, pa, pb, 1, -1, -1) == got, mode
        with switched_context(PWA_PIPE_RUNS="6", PWA_NO_PIPELINE="1") as c:
            assert c.scores(mode, seqs, pa, pb, 1, -1, -1) == got, mode
        rs = np.random.RandomState(5)
        for k in list(rs.choice(len(pa), 40, replace=False)) + [0, n - 1, n, n + 2]:
            assert got[k] == O.score(mode, seqs[pa[k]], seqs[pb[k]], 1, -1, -1)[0], (mode, k)


@pytest.mark.parametrize("engine,n_class", [("stripes", (1, 63, 64)), ("stripes", (65, 127, 128)), ("stripes", (129, 200, 256)),
                                            ("stripes", (257, 300, 511, 512, 513)), ("stripes", (700, 1025, 1100, 1537)),
                                            ("mini", (1, 15, 16, 17, 63, 64)), ("mini", (65, 95, 96, 97, 127, 128)),
                                            ("mini", (129, 150, 159, 160, 161)), ("mini", (191, 192, 193, 255, 256, 257)),
                                            ("wide", (257, 300, 384, 385, 511, 512, 513)), ("wide", (700, 768, 769, 1023, 1024, 1025))])
def test_pair_engine_shapes_around_every_boundary(engine, n_class):
    """The written-out fill chunks (r02): pattern lengths around stripe / workgroup boundaries x text lengths around hand-off chunks
    (16 steps), the lane ramp (63 steps) and the LDS ring (512 columns), for both modes, table and compare scoring, gap-shifted and
    plain global form, with and without the score band -- every op list, end and start cell against the oracle.  "stripes": all
    pairs on the stripe engine (PWA_TB_ENGINE=0; one n-class = one geometry).  "mini" (r03): pattern lengths around every row
    class of the mini-stripe engine (16 RL rows, RL = 4 .. 16) and its 15-step lane ramp; the pairs of one call then fall into
    several geometry classes -- several launches -- and the longest (257) onto the stripe engine."""
    rng = random.Random(sum(n_class) * 7 + len(n_class))
    ms = [1, 2, 15, 16, 17, 31, 47, 48, 49, 62, 63, 64, 65, 79, 80, 81, 95, 127, 128, 129, 191, 255, 256, 257, 511, 512, 513, 520, 1030]
    seqs, pa, pb = [], [], []
    for n in n_class:
        for m in ms:
            p = bytes(rng.choice(b"ACGT") for _ in range(n))
            t = _mutate(rng, (p * (m // max(n, 1) + 2))[:m + 20], 0.1)[:m] if rng.random() < 0.5 else bytes(rng.choice(b"ACGT") for _ in range(m))
            t = (t + bytes(rng.choice(b"ACGT") for _ in range(m)))[:m]
            seqs += [p, t]
            pa.append(len(seqs) - 2)
            pb.append(len(seqs) - 1)
    variants = [({}, False), ({"PWA_NO_GAP_SHIFT": "1"}, False), ({"PWA_NO_PAIR_TABLE": "1"}, False), ({}, True)]
    for env, band in variants:
        if engine == "stripes":
            env = dict(env, PWA_TB_ENGINE="0")
        if engine == "wide":   # one pair per wave for 257 .. 1024 rows, however few (the longest of each class: the stripe engine)
            env = dict(env, PWA_TB_ENGINE="2")
        with switched_context(**env) as c:
            c.set_score_band(band)
            for mode in ("nw", "sw"):
                for sc in [(1, -1, -1), (2, -3, -5)]:
                    res = c.align_batch(mode, seqs, pa, pb, *sc)
                    for k, r in enumerate(res):
                        want = O.align(mode, seqs[pa[k]], seqs[pb[k]], *sc, compact=True)
                        assert (r["score"], r["ops"], tuple(r["end"]), tuple(r["start"])) == \
                            (want["score"], want["ops"], tuple(want["end"]), tuple(want["start"])), (env, band, mode, sc, len(seqs[pa[k]]), len(seqs[pb[k]]))


def test_every_pair_gets_its_own_geometry(ctx):
    """r03: pwa_align_batch / pwa_overlaps choose the band geometry pair by pair -- one 40 kb pattern in a list
    of 100-row patterns no longer turns every short pair into a 256-row stripe.  One call with a 40 000-row pattern (stripe engine,
    RL = 4), 600-row ones (RL = 2), and 2000 patterns of ~100 rows (mini-stripe engine, two RL classes), interleaved: op lists,
    end / start cells, scores and overlaps against the oracle, in caller order."""
    rng = random.Random(40000)
    big_p = O.gen(40, 0, 0, 40000)
    big_t = _mutate(rng, big_p[5000:5900], 0.1)
    seqs, pa, pb = [], [], []

    def add(p, t):
        seqs.extend([p, t])
        pa.append(len(seqs) - 2)
        pb.append(len(seqs) - 1)

    for k in range(2000):
        p = O.gen(41, 0, k, rng.randint(60, 130))
        add(p, _mutate(rng, p, 0.15) + O.gen(41, 1, k, rng.randint(0, 300)))
        if k == 700:
            add(big_p, big_t)
        if k % 400 == 7:
            p6 = O.gen(42, 0, k, 600 + k % 50)
            add(p6, _mutate(rng, p6, 0.1))
    for mode in ("nw", "sw"):
        res = ctx.align_batch(mode, seqs, pa, pb, 1, -1, -1)
        scores, ovl = ctx.overlaps(mode, seqs, pa, pb, 1, -1, -1)
        for k, r in enumerate(res):
            want = O.align(mode, seqs[pa[k]], seqs[pb[k]], 1, -1, -1, compact=True)
            assert (r["score"], r["ops"], tuple(r["end"]), tuple(r["start"])) == \
                (want["score"], want["ops"], tuple(want["end"]), tuple(want["start"])), (mode, k, len(seqs[pa[k]]), len(seqs[pb[k]]))
            assert (scores[k], ovl[k]) == (want["score"], want["overlap"]), (mode, k)


def test_batches_of_mid_sized_patterns_run_one_pair_per_wave(ctx):
    """r03: 300 pairs with patterns of 257 .. 1024 rows in one call -- enough of them for the one-pair-per-wave form of the mini-stripe
    kernels (64 lanes x 8 | 16 rows, a single stripe per pair) -- next to a few short and a few longer ones: op lists, cells, scores and
    overlaps against the oracle."""
    rng = random.Random(1024)
    seqs, pa, pb = [], [], []
    for k in range(300):
        n = rng.choice([257, 300, 400, 511, 512, 513, 700, 1000, 1023, 1024])
        p = O.gen(60, 0, k, n)
        t = (_mutate(rng, p, 0.12) if k % 3 else O.gen(60, 1, k, rng.randint(50, 400)))[:rng.randint(40, 1100)]
        seqs.extend([p, t])
        pa.append(len(seqs) - 2)
        pb.append(len(seqs) - 1)
    for n in (100, 1025, 2000):
        seqs.extend([O.gen(61, 0, n, n), O.gen(61, 1, n, 300)])
        pa.append(len(seqs) - 2)
        pb.append(len(seqs) - 1)
    for mode in ("nw", "sw"):
        res = ctx.align_batch(mode, seqs, pa, pb, 2, -3, -5)
        scores, ovl = ctx.overlaps(mode, seqs, pa, pb, 2, -3, -5)
        for k, r in enumerate(res):
            want = O.align(mode, seqs[pa[k]], seqs[pb[k]], 2, -3, -5, compact=True)
            assert (r["score"], r["ops"], tuple(r["end"]), tuple(r["start"])) == \
                (want["score"], want["ops"], tuple(want["end"]), tuple(want["start"])), (mode, k, len(seqs[pa[k]]), len(seqs[pb[k]]))
            assert (scores[k], ovl[k]) == (want["score"], want["overlap"]), (mode, k)


def test_lists_of_long_patterns_take_taller_stripes(ctx):
    """r03: a list whose multi-stripe pairs fill the chip by themselves runs the stripe engine at RL = 4 (one pair alone: RL = 2, more
    waves in flight).  120 pairs with patterns of 1100 .. 3500 rows (a mutated copy of the pattern's head as text, so that the path
    crosses stripes), a short and a very short one next to them: op lists, cells, scores and overlaps against the oracle."""
    rng = random.Random(3500)
    seqs, pa, pb = [], [], []
    for k in range(120):   # >= 1080 stripes of 128 rows
        n = rng.choice([1100, 2047, 2048, 2049, 3000, 3500])
        p = O.gen(62, 0, k, n)
        t = _mutate(rng, p, 0.1)[:rng.randint(200, 600)] if k % 4 else O.gen(62, 1, k, 300)
        seqs.extend([p, t])
        pa.append(len(seqs) - 2)
        pb.append(len(seqs) - 1)
    for n in (5, 150):
        seqs.extend([O.gen(63, 0, n, n), O.gen(63, 1, n, 200)])
        pa.append(len(seqs) - 2)
        pb.append(len(seqs) - 1)
    for mode in ("nw", "sw"):
        res = ctx.align_batch(mode, seqs, pa, pb, 2, -3, -5)
        scores, ovl = ctx.overlaps(mode, seqs, pa, pb, 2, -3, -5)
        for k, r in enumerate(res):
            want = O.align(mode, seqs[pa[k]], seqs[pb[k]], 2, -3, -5, compact=True)
            assert (r["score"], r["ops"], tuple(r["end"]), tuple(r["start"])) == \
                (want["score"], want["ops"], tuple(want["end"]), tuple(want["start"])), (mode, k, len(seqs[pa[k]]), len(seqs[pb[k]]))
            assert (scores[k], ovl[k]) == (want["score"], want["overlap"]), (mode, k)


@pytest.mark.parametrize("engine", ["auto", "one pair per wave", "stripes"])
def test_local_alignments_with_many_equal_maxima(engine):
    """Two-letter sequences put the maximum score in many cells; hw2.cpp:225-229 starts the traceback at the FIRST one in row-major order.
    The fills keep per-row records of packed keys (value, then the earlier step of a 16-step chunk) and reduce them over rows, lanes and
    stripes: every class of every engine against the oracle -- op lists, start and end cells."""
    rng = random.Random(225)
    seqs, pa, pb = [], [], []
    for k, n in enumerate([1, 2, 15, 16, 17, 31, 33, 64, 65, 100, 150, 160, 161, 255, 256, 257, 300, 400, 513, 700, 1024, 1025, 1500, 2100] * 3):
        p = bytes(rng.choice(b"AC") for _ in range(n)) if k % 5 else (b"A" * n if k % 2 else b"AC" * (n // 2) + b"A" * (n % 2))
        t = bytes(rng.choice(b"AC") for _ in range(rng.choice([1, 15, 16, 17, 40, 200, 333, 800])))
        seqs.extend([p, t])
        pa.append(len(seqs) - 2)
        pb.append(len(seqs) - 1)
    env = {"one pair per wave": {"PWA_TB_ENGINE": "2"}, "stripes": {"PWA_TB_ENGINE": "0"}}.get(engine, {})
    with switched_context(**env) as c:
        for sc in [(1, -1, -1), (2, -3, -5), (1, 0, 0)]:
            res = c.align_batch("sw", seqs, pa, pb, *sc)
            for k, r in enumerate(res):
                want = O.align("sw", seqs[pa[k]], seqs[pb[k]], *sc, compact=True)
                assert (r["score"], tuple(r["end"]), tuple(r["start"]), r["ops"]) == \
                    (want["score"], tuple(want["end"]), tuple(want["start"]), want["ops"]), (engine, sc, k, len(seqs[pa[k]]), len(seqs[pb[k]]))


def test_arenas_of_a_mebibyte_are_coded_on_the_device(ctx):
    """An arena of >= 1 MiB over a small alphabet goes up as raw bytes and is turned into codes by a kernel behind every piece's copy
    (pwalign.hip, build_arena); a NUL byte inside a sequence keeps the host's table pass (the padding between sequences is zeros).
    Alignments and scores of a 1.2 MB list, with and without a NUL, against the oracle on a sample."""
    rng = random.Random(1 << 20)
    for with_nul in (False, True):
        seqs, pa, pb = [], [], []
        for k in range(640):
            p = O.gen(70, 0, k, rng.choice([150, 900, 1500]))
            t = _mutate(rng, p, 0.1)[:rng.randint(100, 1200)] + O.gen(70, 1, k, rng.randint(50, 900))
            seqs.extend([p, t])
            pa.append(len(seqs) - 2)
            pb.append(len(seqs) - 1)
        if with_nul:
            seqs[5] = seqs[5][:40] + b"\x00" + seqs[5][41:]
        assert sum((len(x) + 16) // 16 * 16 for x in seqs) > (1 << 20)
        sample = sorted(rng.sample(range(640), 36) + [2])   # (pair 2 holds the NUL)
        for mode in ("nw", "sw"):
            res = ctx.align_batch(mode, seqs, pa, pb, 1, -1, -1)
            got = ctx.scores(mode, seqs, pa, pb, 1, -1, -1)
            assert got == [r["score"] for r in res]
            for k in sample:
                want = O.align(mode, seqs[pa[k]], seqs[pb[k]], 1, -1, -1, compact=True)
                assert (res[k]["score"], res[k]["ops"], tuple(res[k]["end"]), tuple(res[k]["start"])) == \
                    (want["score"], want["ops"], tuple(want["end"]), tuple(want["start"])), (with_nul, mode, k)


def test_align_batch_cut_into_several_ranges():
    """pwa_align_batch / pwa_overlaps on a list whose bands do not fit one range (PWA_RANGE_BYTES: 3 MB here; at full size the free
    HBM decides): ranges of equal pair counts, every class present in several of them, results in caller order."""
    rng = random.Random(8192)
    seqs, pa, pb = [], [], []
    for k in range(260):
        n = rng.choice([1, 20, 64, 65, 100, 150, 151, 256, 300, 700])
        p = O.gen(64, 0, k, n)
        t = _mutate(rng, p, 0.15)[:rng.randint(1, 400)] if k % 3 else O.gen(64, 1, k, rng.randint(1, 500))
        seqs.extend([p, t])
        pa.append(len(seqs) - 2)
        pb.append(len(seqs) - 1)
    seqs.append(b"")
    pa.append(len(seqs) - 1)
    pb.append(0)
    with switched_context(PWA_RANGE_BYTES=str(3 << 20)) as c, switched_context() as whole:
        for mode in ("nw", "sw"):
            res = c.align_batch(mode, seqs, pa, pb, 2, -3, -5)
            assert res == whole.align_batch(mode, seqs, pa, pb, 2, -3, -5)
            assert c.overlaps(mode, seqs, pa, pb, 2, -3, -5) == whole.overlaps(mode, seqs, pa, pb, 2, -3, -5)
            for k, r in enumerate(res):
                want = O.align(mode, seqs[pa[k]], seqs[pb[k]], 2, -3, -5, compact=True)
                assert (r["score"], r["ops"], tuple(r["end"]), tuple(r["start"])) == \
                    (want["score"], want["ops"], tuple(want["end"]), tuple(want["start"])), (mode, k)


@pytest.mark.parametrize("how", ["padded regions", "PWA_NO_TILED_OPS"])
def test_op_lists_through_the_staging_copy(how):
    """pwa_align_batch with op regions that do NOT follow one another without a gap (a C caller with aligned regions), and with the
    tiled fast path switched off: the op lists come back through the staging copy and must equal the tiled result and the oracle.
    (Bytes of a region beyond n_ops[k] are undefined, include/pwalign.h.)"""
    rng = random.Random(1813)
    seqs = [bytes(rng.choice(b"ACGT") for _ in range(rng.choice([0, 1, 50, 150, 300, 700]))) for _ in range(40)]
    pa = [rng.randrange(40) for _ in range(120)]
    pb = [rng.randrange(40) for _ in range(120)]
    with switched_context(**({"PWA_NO_TILED_OPS": "1"} if how != "padded regions" else {})) as c:
        for mode in ("nw", "sw"):
            res = c.align_batch(mode, seqs, pa, pb, 2, -3, -5, region_pad=16 if how == "padded regions" else 0)
            for k, r in enumerate(res):
                want = O.align(mode, seqs[pa[k]], seqs[pb[k]], 2, -3, -5, compact=True)
                assert (r["score"], r["ops"], tuple(r["end"]), tuple(r["start"])) == \
                    (want["score"], want["ops"], tuple(want["end"]), tuple(want["start"])), (how, mode, k)


def _indel_blocks(rng, s, n_events, max_len):
    """s with n_events blocks of up to max_len symbols cut out or spliced in: long runs of 'u' / 'l' in the optimal path."""
    out = bytearray(s)
    for _ in range(n_events):
        at = rng.randrange(0, max(1, len(out)))
        ln = rng.randint(1, max_len)
        if rng.random() < 0.5:
            del out[at:at + ln]
        else:
            out[at:at] = bytes(rng.choice(b"ACGT") for _ in range(ln))
    return bytes(out)


@pytest.mark.parametrize("force_rl", [None, "4", "2"])
def test_walk_across_gap_runs_drift_and_ties(force_rl):
    """The op-list walk (r02) follows the path over seven diagonals per LDS round trip and ends a trip when the path drifts off them,
    leaves the staged windows or the stripe: paths with long runs of gaps in one direction, zig-zags of single gaps, tie-riddled
    homopolymers and scorings that prefer gaps, over one- and many-stripe patterns -- every op list against the oracle."""
    rng = random.Random(77)
    seqs, pa, pb = [], [], []

    def add(p, t):
        seqs.extend([p, t])
        pa.append(len(seqs) - 2)
        pb.append(len(seqs) - 1)

    for n in (90, 200, 700, 1500):
        base = bytes(rng.choice(b"ACGT") for _ in range(n))
        add(base, _indel_blocks(rng, base, 6, 5))            # short gap runs: the path wanders over a few diagonals
        add(base, _indel_blocks(rng, base, 5, 60))           # long runs: off the seven diagonals at once
        add(_indel_blocks(rng, base, 4, 150), base)
        add(base, base[n // 3:] + base[:n // 3])             # a rotation: two long gap runs at the ends
        zig = bytearray()
        for k, c in enumerate(base):                         # single-symbol indels every few columns
            if k % 7 == 3:
                continue
            zig.append(c)
            if k % 5 == 1:
                zig.append(rng.choice(b"ACGT"))
        add(base, bytes(zig))
        add(b"A" * n, b"A" * (n + 37))                       # ties everywhere: the reference's order of preference decides
        add(b"AC" * (n // 2), b"CA" * (n // 2 + 11))
        add(base, bytes(rng.choice(b"ACGT") for _ in range(n // 2 + 5)))
    with switched_context(**({"PWA_FORCE_RL": force_rl} if force_rl else {})) as c:
        for mode in ("nw", "sw"):
            for sc in [(1, -1, -1), (1, -3, -1), (2, -1, -3), (0, 0, 0), (1, 1, 1), (-1, 2, 1), (1, -1, 0)]:
                res = c.align_batch(mode, seqs, pa, pb, *sc)
                for k, r in enumerate(res):
                    want = O.align(mode, seqs[pa[k]], seqs[pb[k]], *sc, compact=True)
                    assert (r["score"], r["ops"], tuple(r["end"]), tuple(r["start"])) == \
                        (want["score"], want["ops"], tuple(want["end"]), tuple(want["start"])), (force_rl, mode, sc, k, len(seqs[pa[k]]), len(seqs[pb[k]]))


def test_pipeline_handoff_under_uneven_concurrent_load(ctx):
    """Many multi-super-stripe pairs of very different shapes in ONE launch: every inter-workgroup hand-off
    (helper wave, sc1 rows + progress counters) runs while other workgroups stream bands at different rates.
    Every op of every pair must equal the oracle's."""
    rng = random.Random(2024)
    seqs = []
    for _ in range(40):
        n = rng.choice([600, 700, 1030, 1500, 2100, 3000])
        seqs.append(bytes(rng.choice(b"ACGT") for _ in range(n + rng.randint(0, 40))))
    for _ in range(12):
        seqs.append(bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 300))))
    pa = [rng.randrange(40) for _ in range(70)] + [40 + rng.randrange(12) for _ in range(10)]
    pb = [rng.randrange(52) for _ in range(80)]
    for mode in ("nw", "sw"):
        res = ctx.align_batch(mode, seqs, pa, pb, 1, -1, -1)
        for k, r in enumerate(res):
            want = O.align(mode, seqs[pa[k]], seqs[pb[k]], 1, -1, -1, compact=True)
            assert r["score"] == want["score"], (mode, k)
            assert r["ops"] == want["ops"], (mode, k, len(seqs[pa[k]]), len(seqs[pb[k]]))
            assert tuple(r["end"]) == tuple(want["end"]) and tuple(r["start"]) == tuple(want["start"])
