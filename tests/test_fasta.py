"""FASTA ingest: pwa_fasta_read against the reference's readFasta (hw2.cpp:25-57).

Expected outputs come from the compiled reference function (tests/golden/fasta.json, generator committed next
to it); the oracle restatement and -- in the dev container -- the reference itself are checked on the same
inputs.  No GPU involved."""
import base64
import json
import os
import random

import pytest

import oracle_lib as O
from conftest import GOLDEN, load_pkg

CASES = json.load(open(os.path.join(GOLDEN, "fasta.json")))


def _seqs(blob, off, lo, hi):
    return [blob[off[k]:off[k + 1]] for k in range(lo, hi)]


@pytest.fixture(scope="module")
def pkg():
    return load_pkg()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_read_fasta_matches_reference_vectors(case, tmp_path):
    path = tmp_path / "in.fasta"
    path.write_bytes(base64.b64decode(case["content_b64"]))
    want = [base64.b64decode(s) for s in case["sequences_b64"]]
    assert O.read_fasta(str(path)) == want
    if O.have_ref():
        assert O.ref_read_fasta(str(path)) == want


@pytest.mark.parametrize("min_chunk", [None, 1, 7, 64, 1000])
def test_fasta_reader_matches_reference_vectors(pkg, tmp_path, monkeypatch, min_chunk):
    """every golden case, one file at a time and all files in one call; tiny chunk sizes force records, lines and
    blank runs to straddle the per-thread chunks."""
    if min_chunk is not None:
        monkeypatch.setenv("PWA_FASTA_MIN_CHUNK", str(min_chunk))
    paths, wants = [], []
    for c in CASES:
        p = tmp_path / (c["name"] + ".fasta")
        p.write_bytes(base64.b64decode(c["content_b64"]))
        paths.append(str(p))
        wants.append([base64.b64decode(s) for s in c["sequences_b64"]])
    for threads in (1, 2, 3, 8):
        for p, want in zip(paths, wants):
            blob, off, first = pkg.read_fasta(p, threads)
            assert first == [0, len(want)]
            assert _seqs(blob, off, 0, len(want)) == want, (p, threads)
        blob, off, first = pkg.read_fasta(paths, threads)
        assert len(first) == len(paths) + 1 and first[0] == 0
        for i, want in enumerate(wants):
            assert _seqs(blob, off, first[i], first[i + 1]) == want, (paths[i], threads)


def test_fasta_reader_large_random_file_against_oracle(pkg, tmp_path):
    rng = random.Random(12)
    parts = []
    for r in range(3000):
        parts.append(b">r%d\n" % r if rng.random() < 0.9 else b">\r\n")
        for _ in range(rng.randint(0, 12)):
            body = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 120)))
            parts.append(body + rng.choice([b"\n", b"\r\n", b" \n", b"\n\n"]))
    content = b"".join(parts)
    p = tmp_path / "big.fasta"
    p.write_bytes(content)
    want = O.read_fasta(str(p))
    os.environ["PWA_FASTA_MIN_CHUNK"] = "65536"
    try:
        for threads in (1, 4, 16):
            blob, off, first = pkg.read_fasta(str(p), threads)
            assert _seqs(blob, off, 0, first[1]) == want
    finally:
        del os.environ["PWA_FASTA_MIN_CHUNK"]


def test_fasta_reader_errors(pkg, tmp_path):
    good = tmp_path / "a.fasta"
    good.write_bytes(b">a\nAC\n")
    with pytest.raises(pkg.PwaError, match="cannot open"):
        pkg.read_fasta(str(tmp_path / "missing.fasta"))
    with pytest.raises(pkg.PwaError, match="missing2"):      # the index of the failing file comes back
        pkg.read_fasta([str(good), str(tmp_path / "missing2.fasta")])
    with pytest.raises(pkg.PwaError):
        pkg.read_fasta(str(tmp_path))                          # a directory
    assert pkg.read_fasta([]) == (b"", [0], [0])
    assert O.read_fasta(str(tmp_path / "missing.fasta")) is None
