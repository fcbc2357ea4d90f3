"""ctypes access to the CHECKER (oracle/liboracle_hw2.so) and, in the dev container only, to the
unmodified reference built into oracle/_ref/libhw2_ref.so.

Test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle_hw2.so")
ORACLE_CLI = os.path.join(ORACLE_DIR, "hw2_oracle_cli")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libhw2_ref.so")
REF_CLI = os.path.join(ORACLE_DIR, "_ref", "hw2_ref")
ORACLE3_SO = os.path.join(ORACLE_DIR, "liboracle_hw3.so")
ORACLE4_SO = os.path.join(ORACLE_DIR, "liboracle_hw4.so")
ORACLE4_CLI = os.path.join(ORACLE_DIR, "hw4_oracle_cli")
REF4_SO = os.path.join(ORACLE_DIR, "_ref", "libhw4_ref.so")
REF4_CLI = os.path.join(ORACLE_DIR, "_ref", "hw4_ref")
REF3_SO = os.path.join(ORACLE_DIR, "_ref", "libhw3_ref.so")
REF3_CLI = os.path.join(ORACLE_DIR, "_ref", "hw3_ref")


class _OrcResult(C.Structure):
    _fields_ = [
        ("score", C.c_int32),
        ("aligned_pattern", C.c_void_p),
        ("aligned_reference", C.c_void_p),
        ("cigar", C.c_void_p),
        ("mdz", C.c_void_p),
        ("ops", C.c_void_p),
        ("n_ops", C.c_size_t),
        ("end_i", C.c_size_t),
        ("end_j", C.c_size_t),
        ("start_i", C.c_size_t),
        ("start_j", C.c_size_t),
    ]


class _RefResult(C.Structure):
    _fields_ = [
        ("score", C.c_int),
        ("aligned_pattern", C.c_void_p),
        ("aligned_reference", C.c_void_p),
        ("cigar", C.c_void_p),
        ("mdz", C.c_void_p),
    ]


def build_oracle(with_ref=True):
    """Compile the checker (and oracle/_ref when the reference checkout is present)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True)
    if with_ref:
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, "ref"], check=True)


_lib = None
_ref = None


def oracle():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle(with_ref=False)
        lib = C.CDLL(ORACLE_SO)
        sig = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int]
        for name in ("orc_nw", "orc_sw", "orc_nw_compact", "orc_sw_compact"):
            f = getattr(lib, name)
            f.argtypes = sig
            f.restype = C.POINTER(_OrcResult)
        lib.orc_free.argtypes = [C.POINTER(_OrcResult)]
        lib.orc_free.restype = None
        lib.orc_nw_score.argtypes = sig
        lib.orc_nw_score.restype = C.c_int32
        lib.orc_sw_score.argtypes = sig + [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        lib.orc_sw_score.restype = C.c_int32
        lib.orc_overlap.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        lib.orc_overlap.restype = C.c_int
        lib.orc_cigar.argtypes = [C.c_char_p, C.c_size_t]
        lib.orc_cigar.restype = C.c_void_p
        lib.orc_mdz.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_size_t]
        lib.orc_mdz.restype = C.c_void_p
        lib.orc_matrices.argtypes = [C.c_int] + sig + [C.c_void_p, C.c_void_p]
        lib.orc_matrices.restype = None
        lib.orc_gen.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_size_t, C.c_char_p]
        lib.orc_gen.restype = None
        lib.orc_hw2_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        lib.orc_hw2_main.restype = C.c_int
        lib.orc_read_fasta.argtypes = [C.c_char_p, C.POINTER(_OrcFasta)]
        lib.orc_read_fasta.restype = C.c_int
        lib.orc_free_fasta.argtypes = [C.POINTER(_OrcFasta)]
        lib.orc_free_fasta.restype = None
        _lib = lib
    return _lib


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        lib = C.CDLL(REF_SO)
        sig = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int]
        for name in ("ref_nw", "ref_sw"):
            f = getattr(lib, name)
            f.argtypes = sig
            f.restype = C.POINTER(_RefResult)
        lib.ref_free.argtypes = [C.POINTER(_RefResult)]
        lib.ref_overlap.argtypes = [C.c_char_p, C.c_char_p]
        lib.ref_overlap.restype = C.c_int
        lib.ref_read_fasta.argtypes = [C.c_char_p]
        lib.ref_read_fasta.restype = C.POINTER(_RefFasta)
        lib.ref_free_fasta.argtypes = [C.POINTER(_RefFasta)]
        lib.ref_free_fasta.restype = None
        _ref = lib
    return _ref


class _OrcFasta(C.Structure):   # oracle/hw2_oracle.h: orc_fasta
    _fields_ = [("count", C.c_size_t), ("seq", C.POINTER(C.c_void_p)), ("len", C.POINTER(C.c_size_t))]


class _RefFasta(C.Structure):   # oracle/ref_shim.cpp: ref_fasta
    _fields_ = [("count", C.c_size_t), ("blob", C.c_void_p), ("off", C.POINTER(C.c_size_t))]


def read_fasta(path):
    """Oracle readFasta (hw2.cpp:25-57) -> list of bytes, or None when the file cannot be opened."""
    lib = oracle()
    f = _OrcFasta()
    if lib.orc_read_fasta(os.fsencode(path), C.byref(f)) != 0:
        return None
    out = [C.string_at(f.seq[i], f.len[i]) for i in range(f.count)]
    lib.orc_free_fasta(C.byref(f))
    return out


def ref_read_fasta(path):
    """The compiled reference's readFasta; existing files only (it exit(1)s otherwise)."""
    assert os.path.exists(path)
    lib = ref()
    f = lib.ref_read_fasta(os.fsencode(path)).contents
    out = [C.string_at(f.blob + f.off[i], f.off[i + 1] - f.off[i]) for i in range(f.count)]
    lib.ref_free_fasta(C.byref(f))
    return out


def _s(ptr):
    return C.string_at(ptr) if ptr else b""


def _as_bytes(x):
    return x if isinstance(x, (bytes, bytearray)) else x.encode()


def align(mode, p, t, match, mismatch, gap, compact=False):
    """Oracle alignment -> dict. mode 'nw' | 'sw'."""
    p, t = _as_bytes(p), _as_bytes(t)
    lib = oracle()
    fn = getattr(lib, "orc_%s%s" % (mode, "_compact" if compact else ""))
    r = fn(p, len(p), t, len(t), match, mismatch, gap)
    if not r:
        raise MemoryError("oracle allocation failed")
    o = r.contents
    out = dict(
        score=o.score,
        aligned_pattern=_s(o.aligned_pattern),
        aligned_reference=_s(o.aligned_reference),
        cigar=_s(o.cigar),
        mdz=_s(o.mdz),
        ops=C.string_at(o.ops, o.n_ops),
        end=(o.end_i, o.end_j),
        start=(o.start_i, o.start_j),
    )
    out["overlap"] = lib.orc_overlap(out["aligned_pattern"], out["aligned_reference"], len(out["aligned_pattern"]))
    lib.orc_free(r)
    return out


def matrices(mode, p, t, match, mismatch, gap):
    """The reference's dp (int32) and traceback (char) matrices, (n+1, m+1) numpy arrays."""
    import numpy as np
    p, t = _as_bytes(p), _as_bytes(t)
    dp = np.zeros((len(p) + 1, len(t) + 1), dtype=np.int32)
    tb = np.zeros((len(p) + 1, len(t) + 1), dtype=np.uint8)
    oracle().orc_matrices(0 if mode == "nw" else 1, p, len(p), t, len(t), match, mismatch, gap,
                          dp.ctypes.data_as(C.c_void_p), tb.ctypes.data_as(C.c_void_p))
    return dp, tb


def score(mode, p, t, match, mismatch, gap):
    """Oracle score-only. Returns (score, end_i, end_j); NW end = (n, m)."""
    p, t = _as_bytes(p), _as_bytes(t)
    lib = oracle()
    if mode == "nw":
        return lib.orc_nw_score(p, len(p), t, len(t), match, mismatch, gap), len(p), len(t)
    ei, ej = C.c_size_t(0), C.c_size_t(0)
    s = lib.orc_sw_score(p, len(p), t, len(t), match, mismatch, gap, C.byref(ei), C.byref(ej))
    return s, ei.value, ej.value


def ref_align(mode, p, t, match, mismatch, gap):
    """The unmodified reference function (dev container only)."""
    p, t = _as_bytes(p), _as_bytes(t)
    lib = ref()
    fn = lib.ref_nw if mode == "nw" else lib.ref_sw
    r = fn(p, len(p), t, len(t), match, mismatch, gap)
    o = r.contents
    out = dict(
        score=o.score,
        aligned_pattern=_s(o.aligned_pattern),
        aligned_reference=_s(o.aligned_reference),
        cigar=_s(o.cigar),
        mdz=_s(o.mdz),
    )
    out["overlap"] = lib.ref_overlap(out["aligned_pattern"], out["aligned_reference"])
    lib.ref_free(r)
    return out


def gen(seed, stream, ident, length):
    """SURVEY.md 8(d) generator."""
    buf = C.create_string_buffer(length + 1)
    oracle().orc_gen(seed, stream, ident, length, buf)
    return buf.raw[:length]


def run_cli(exe, args, cwd=None):
    """Run a hw2-compatible CLI; returns (rc, stderr bytes)."""
    pr = subprocess.run([exe] + [str(a) for a in args], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return pr.returncode, pr.stderr


# ---------------------------------------------------------------------------- hw3 affine score pass
_lib3 = None
_ref3 = None
_SIG3 = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int]


def oracle3():
    global _lib3
    if _lib3 is None:
        if not os.path.exists(ORACLE3_SO):
            build_oracle(with_ref=False)
        lib = C.CDLL(ORACLE3_SO)
        lib.orc3_affine_score.argtypes = _SIG3
        lib.orc3_affine_score.restype = C.c_int32
        lib.orc3_center.argtypes = [C.POINTER(C.c_int32), C.c_size_t, C.POINTER(C.c_int64)]
        lib.orc3_center.restype = C.c_size_t
        lib.orc3_affine_align.argtypes = _SIG3
        lib.orc3_affine_align.restype = C.POINTER(_Orc3Alignment)
        lib.orc3_free_alignment.argtypes = [C.POINTER(_Orc3Alignment)]
        lib.orc3_free_alignment.restype = None
        lib.orc3_hw3_main.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        lib.orc3_hw3_main.restype = C.c_int
        _lib3 = lib
    return _lib3


class _Orc3Alignment(C.Structure):   # oracle/hw3_oracle.h: orc3_alignment
    _fields_ = [("score", C.c_int32), ("a1", C.c_void_p), ("a2", C.c_void_p), ("ops", C.c_void_p), ("len", C.c_size_t)]


def affine_align(s1, s2, match, mismatch, go, ge):
    """Oracle: hw3.cpp:23-135 with strings -> dict(score, a1, a2, ops); ops in traceback order ('M'/'D'/'I')."""
    s1, s2 = _as_bytes(s1), _as_bytes(s2)
    lib = oracle3()
    r = lib.orc3_affine_align(s1, len(s1), s2, len(s2), match, mismatch, go, ge)
    if not r:
        raise MemoryError("oracle allocation failed")
    o = r.contents
    out = dict(score=o.score, a1=C.string_at(o.a1, o.len), a2=C.string_at(o.a2, o.len), ops=C.string_at(o.ops, o.len))
    lib.orc3_free_alignment(r)
    return out


def ref_affine_align(s1, s2, match, mismatch, go, ge):
    """The unmodified hw3.cpp affine_alignment with the strings requested (dev container only)."""
    ref_affine_score(b"A", b"A", 1, -1, -1, -1)   # loads _ref3
    f = _ref3.ref3_affine_align
    f.argtypes = _SIG3 + [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    f.restype = C.c_int
    _ref3.ref3_free.argtypes = [C.c_void_p]
    _ref3.ref3_free.restype = None
    s1, s2 = _as_bytes(s1), _as_bytes(s2)
    p1, p2 = C.c_void_p(), C.c_void_p()
    score = f(s1, len(s1), s2, len(s2), match, mismatch, go, ge, C.byref(p1), C.byref(p2))
    out = dict(score=score, a1=C.string_at(p1), a2=C.string_at(p2))
    _ref3.ref3_free(p1)
    _ref3.ref3_free(p2)
    return out


ORACLE3_CLI = os.path.join(os.path.dirname(ORACLE3_SO), "hw3_oracle_cli")
REF3_CLI = os.path.join(os.path.dirname(REF3_SO), "hw3_ref")


def have_ref3():
    return os.path.exists(REF3_SO)


def affine_score(s1, s2, match, mismatch, go, ge):
    """Oracle: hw3.cpp affine_alignment score (max of V, F, E at (n, m))."""
    s1, s2 = _as_bytes(s1), _as_bytes(s2)
    return oracle3().orc3_affine_score(s1, len(s1), s2, len(s2), match, mismatch, go, ge)


def ref_affine_score(s1, s2, match, mismatch, go, ge):
    """The unmodified hw3.cpp function (dev container only)."""
    global _ref3
    if _ref3 is None:
        _ref3 = C.CDLL(REF3_SO)
        _ref3.ref3_affine_score.argtypes = _SIG3
        _ref3.ref3_affine_score.restype = C.c_int
    s1, s2 = _as_bytes(s1), _as_bytes(s2)
    return _ref3.ref3_affine_score(s1, len(s1), s2, len(s2), match, mismatch, go, ge)


def center(pair_scores_upper, n_seq):
    """hw3.cpp:230-251: (center index, star scores)."""
    arr = (C.c_int32 * max(len(pair_scores_upper), 1))(*pair_scores_upper)
    sums = (C.c_int64 * n_seq)()
    c = oracle3().orc3_center(arr, n_seq, sums)
    return c, list(sums)


def read_fasta_hw3(path):
    """hw3.cpp:137-167 readFASTA: (header, sequence) records; whitespace inside lines dropped."""
    recs, header, seq = [], None, b""
    for line in open(path, "rb").read().split(b"\n"):
        if not line:
            continue
        if line[:1] == b">":
            if header:
                recs.append((header, seq))
                seq = b""
            header = line[1:]
        else:
            seq += bytes(ch for ch in line if ch not in b" \t\n\v\f\r")
    if header:
        recs.append((header, seq))
    return recs


# ---------------------------------------------------------------------------- hw4 NW distance / UPGMA
_lib4 = None
_ref4 = None


def oracle4():
    global _lib4
    if _lib4 is None:
        if not os.path.exists(ORACLE4_SO):
            build_oracle(with_ref=False)
        lib = C.CDLL(ORACLE4_SO)
        lib.orc4_nw_distance.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32)]
        lib.orc4_nw_distance.restype = C.c_int32
        lib.orc4_upgma.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_char_p), C.c_size_t]
        lib.orc4_upgma.restype = C.c_void_p
        _lib4 = lib
    return _lib4


def have_ref4():
    return os.path.exists(REF4_SO)


def nw_distance(s1, s2, match, mismatch, gap):
    """Oracle: (distance, score) of hw4.cpp's needleman_wunsch + distance rule (16-72, 146-152)."""
    s1, s2 = _as_bytes(s1), _as_bytes(s2)
    sc = C.c_int32(0)
    d = oracle4().orc4_nw_distance(s1, len(s1), s2, len(s2), match, mismatch, gap, C.byref(sc))
    return d, sc.value


def ref_nw_distance(s1, s2, match, mismatch, gap):
    global _ref4
    if _ref4 is None:
        _ref4 = C.CDLL(REF4_SO)
        _ref4.ref4_nw_distance.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int]
        _ref4.ref4_nw_distance.restype = C.c_int
    s1, s2 = _as_bytes(s1), _as_bytes(s2)
    return _ref4.ref4_nw_distance(s1, len(s1), s2, len(s2), match, mismatch, gap)


def upgma(dist_rows, names):
    """Oracle UPGMA (hw4.cpp:162-228): dense symmetric matrix -> Newick bytes."""
    n = len(names)
    flat = (C.c_double * max(n * n, 1))(*[x for row in dist_rows for x in row])
    arr = (C.c_char_p * max(n, 1))(*[_as_bytes(x) for x in names])
    p = oracle4().orc4_upgma(flat, arr, n)
    return C.string_at(p)
