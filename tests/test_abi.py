"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the
header declares, refuses to work without a GPU (no CPU fallback), and its host-only
post-processing (CIGAR / MD:Z / gapped strings / overlap) matches the reference fixtures."""
import ctypes as C
import os
import re
import subprocess

import pytest

import oracle_lib as O
from conftest import B, ROOT, load_golden, load_pkg


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "pwalign.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(pwa_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    pkg = load_pkg()
    L = pkg.lib()
    syms = declared_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(pkg.EXPORTS) == syms
    assert L.pwa_version().decode().endswith("gfx950")


def test_library_has_gfx950_code_object():
    pkg = load_pkg()
    out = subprocess.run(["strings", "-a", pkg.LIB_PATH], stdout=subprocess.PIPE).stdout
    assert b"gfx950" in out and b"batch_scores_kernel" in out and b"pair_fill_kernel" in out


def test_dropin_library_exports_the_reference_signatures():
    """libhw2_dropin.so (INTEGRATION.md Option B): the two functions under the names a C++ caller compiled against
    hw2.cpp:118 / 192 would link to (Itanium mangling of the reference's exact signatures) -- no compute call here."""
    pkg = load_pkg()
    so = os.path.join(os.path.dirname(pkg.LIB_PATH), "libhw2_dropin.so")
    assert os.path.exists(so)
    out = subprocess.run(["nm", "-D", "--defined-only", so], stdout=subprocess.PIPE).stdout.decode()
    sig = "RKNSt7__cxx1112basic_stringIcSt11char_traitsIcESaIcEEES6_iii"   # (const std::string&, const std::string&, int, int, int)
    assert "_Z30globalAlignmentNeedlemanWunsch" + sig in out
    assert "_Z27localAlignmentSmithWaterman" + sig in out
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libhw2_ref.so")
    if os.path.exists(ref_so):   # dev container: the compiled reference defines exactly these two symbols
        ref = subprocess.run(["nm", "-D", "--defined-only", ref_so], stdout=subprocess.PIPE).stdout.decode()
        for name in ("_Z30globalAlignmentNeedlemanWunsch" + sig, "_Z27localAlignmentSmithWaterman" + sig):
            assert name in ref


def test_host_sorting_helpers_selftest():
    """The scheduler's stable sorts (counting sort incl. its multi-threaded form, length sort, radix sort) against std::stable_sort on
    lists of up to 2^20 elements -- inside the library, no GPU involved."""
    L = load_pkg().lib()
    L.pwa_selftest_host.argtypes = [C.c_uint32]
    L.pwa_selftest_host.restype = C.c_int
    for seed in (1, 2, 12345):
        assert L.pwa_selftest_host(seed) == 0


def _no_gpu():
    return not os.path.exists("/dev/kfd")


@pytest.mark.skipif(not _no_gpu(), reason="only meaningful on a machine without a GPU")
def test_no_cpu_fallback_without_gpu():
    pkg = load_pkg()
    with pytest.raises(pkg.PwaError):
        pkg.Context(0)


def test_error_strings():
    L = load_pkg().lib()
    assert L.pwa_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4, -5, -99):
        assert L.pwa_strerror(code)
    assert L.pwa_ctx_create(0, None) == -1


@pytest.mark.parametrize("name", ["bundled", "edge", "random"])
def test_format_alignment_matches_reference(name):
    """ops + end cell (from the oracle's walk) -> strings through the C ABI == reference strings."""
    pkg = load_pkg()
    for rec in load_golden(name):
        p, t = B(rec["p"]), B(rec["t"])
        o = O.align(rec["mode"], p, t, *rec["scoring"])
        f = pkg.format_alignment(p, t, o["ops"], o["end"])
        assert f["cigar"] == B(rec["cigar"])
        assert f["mdz"] == B(rec["mdz"])
        assert f["aligned_pattern"] == B(rec["aligned_pattern"])
        assert f["aligned_reference"] == B(rec["aligned_reference"])
        assert f["overlap"] == rec["overlap"]
        assert pkg.alignment_overlap(p, t, o["ops"], o["end"]) == rec["overlap"]
    assert pkg.alignment_overlap(b"A-C", b"A-C", b"MMM", (3, 3)) == pkg.format_alignment(b"A-C", b"A-C", b"MMM", (3, 3))["overlap"] == 1


def test_format_alignment_rejects_inconsistent_input():
    pkg = load_pkg()
    with pytest.raises(pkg.PwaError):
        pkg.format_alignment(b"AC", b"AC", b"MMM", (2, 2))
    with pytest.raises(pkg.PwaError):
        pkg.format_alignment(b"AC", b"AC", b"MX", (2, 2))


def test_cli_argument_errors_match_reference(tmp_path):
    """Paths of the CLI that never reach the GPU: usage, unreadable input, count mismatch,
    neither -g nor -l (empty output), unwritable output with zero pairs."""
    pkg = load_pkg()
    exe = pkg.CLI_PATH
    assert os.path.exists(exe), "host CLI not built"
    import shutil
    from conftest import GOLDEN
    cli = load_golden("cli")
    for f in ("patterns.fasta", "texts.fasta"):
        shutil.copyfile(os.path.join(GOLDEN, f), tmp_path / f)
    for name, content in cli["files"].items():
        (tmp_path / name).write_bytes(B(content))
    ran = 0
    for case in cli["cases"]:
        needs_gpu = case["rc"] == 0 and case["output"] not in (None, "")
        if needs_gpu or "nodir/out.txt" in case["args"]:
            continue
        outp = tmp_path / "out.txt"
        if outp.exists():
            outp.unlink()
        pr = subprocess.run([exe] + case["args"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert pr.returncode == case["rc"], case["args"]
        assert pr.stderr.replace(exe.encode(), b"hw2") == B(case["stderr"]), case["args"]
        got = outp.read_bytes() if outp.exists() else None
        want = B(case["output"]) if case["output"] is not None else None
        assert got == want, case["args"]
        ran += 1
    assert ran >= 5


def test_pack_sequences_takes_a_tuple_of_three_sequences_as_sequences(pkg):
    """a 3-tuple of byte strings is a list of three sequences, not an already packed (blob, offsets, list) triple"""
    blob, off, seqs = pkg.pack_sequences((b"ACGT", b"ACG", b"AC"))
    assert blob == b"ACGTACGAC" and list(off) == [0, 4, 7, 9] and seqs == [b"ACGT", b"ACG", b"AC"]
    again = pkg.pack_sequences((blob, off, seqs))
    assert again[0] is blob and again[1] is off
