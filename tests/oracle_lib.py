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
        _ref = lib
    return _ref


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
