"""bioinformatics-algorithms_amd -- MI355X-native pairwise alignment (NW / SW, linear gap, int32).

Python is only a thin ctypes binding over the C ABI of ``libpwalign.so`` (include/pwalign.h); all
compute is in hand-written HIP kernels for gfx950.  There is NO CPU fallback: if the shared library
is missing or no MI355X is visible, ``Context()`` raises.

The directory name contains a hyphen, so import it through ``load()`` in ``_loader.py`` (or
``importlib`` with ``spec_from_file_location``); tests/bench do exactly that.

Mirror of the reference interface (Local_Global_Alignment/hw2.cpp):
  ``Context.align('nw'|'sw', pattern, text, match, mismatch, gap)`` returns the five fields of the
  reference's ``AlignmentResult`` (hw2.cpp:17-23) as a dict -- what
  ``globalAlignmentNeedlemanWunsch`` (118) / ``localAlignmentSmithWaterman`` (192) return;
  ``Context.scores(...)`` is the scores-only pass over the pair loop (328-338).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PWA_LIB") or os.path.join(_HERE, "libpwalign.so")   # PWA_LIB: A/B builds in experiments
CLI_PATH = os.path.join(_HERE, "host", "hw2_amd")
HW3_CLI_PATH = os.path.join(_HERE, "host", "hw3_amd")
CLI4_PATH = os.path.join(_HERE, "host", "hw4_amd")

MODE = {"nw": 0, "sw": 1, "global": 0, "local": 1}

EXPORTS = [
    "pwa_version", "pwa_strerror", "pwa_selftest_host", "pwa_ctx_create", "pwa_ctx_destroy", "pwa_last_error", "pwa_ctx_set_score_band", "pwa_scores",
    "pwa_batch_create", "pwa_affine_batch_create", "pwa_scores_affine", "pwa_align_affine_batch", "pwa_nwdist_batch_create", "pwa_distances", "pwa_upgma_newick", "pwa_batch_run", "pwa_batch_d_scores", "pwa_batch_set_d_scores", "pwa_batch_fetch", "pwa_batch_info",
    "pwa_batch_last_ms", "pwa_batch_run_times", "pwa_batch_destroy", "pwa_align", "pwa_align_matrices", "pwa_align_last_stats", "pwa_align_batch", "pwa_overlaps",
    "pwa_cigar_bound", "pwa_mdz_bound", "pwa_format_alignment", "pwa_alignment_overlap",
    "pwa_fasta_read", "pwa_fasta_n_seq", "pwa_fasta_bytes", "pwa_fasta_offsets", "pwa_fasta_first_seq", "pwa_fasta_free",
]


class PwaError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libpwalign.so (built in-tree by __graft_entry__.build() / csrc/Makefile). Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PwaError("libpwalign.so is not built (%s); run `make -C bioinformatics-algorithms_amd/csrc`: "
                       "there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u8p, u32p, u64p, i32p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)
    L.pwa_version.restype = C.c_char_p
    L.pwa_strerror.restype = C.c_char_p
    L.pwa_strerror.argtypes = [C.c_int]
    L.pwa_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.pwa_ctx_destroy.argtypes = [vp]
    L.pwa_ctx_destroy.restype = None
    L.pwa_last_error.argtypes = [vp]
    L.pwa_last_error.restype = C.c_char_p
    L.pwa_ctx_set_score_band.argtypes = [vp, C.c_int]
    batch_in = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, u64p, C.c_uint32, u32p, u32p, C.c_uint64]
    L.pwa_scores.argtypes = batch_in + [i32p, u32p, u32p]
    L.pwa_batch_create.argtypes = batch_in + [C.c_int, C.POINTER(vp)]
    affine_in = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, u64p, C.c_uint32, u32p, u32p, C.c_uint64]
    L.pwa_affine_batch_create.argtypes = affine_in + [C.POINTER(vp)]
    L.pwa_scores_affine.argtypes = affine_in + [i32p]
    L.pwa_nwdist_batch_create.argtypes = batch_in[:1] + batch_in[2:] + [C.POINTER(vp)]
    L.pwa_distances.argtypes = batch_in[:1] + batch_in[2:] + [i32p]
    L.pwa_upgma_newick.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_char_p), C.c_uint32, vp, C.c_uint64, u64p]
    L.pwa_batch_run.argtypes = [vp, vp]
    L.pwa_batch_d_scores.argtypes = [vp]
    L.pwa_batch_d_scores.restype = vp
    L.pwa_batch_set_d_scores.argtypes = [vp, vp]
    L.pwa_batch_fetch.argtypes = [vp, i32p, u32p, u32p]
    L.pwa_batch_info.argtypes = [vp, u64p, u64p, u64p, C.POINTER(C.c_char_p)]
    L.pwa_batch_last_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.pwa_batch_run_times.argtypes = [vp, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
    L.pwa_batch_destroy.argtypes = [vp]
    L.pwa_batch_destroy.restype = None
    L.pwa_align.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_uint64, vp, C.c_uint64, i32p, vp,
                            C.c_uint64, u64p, u64p, u64p]
    L.pwa_align_matrices.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_uint64, vp, C.c_uint64, vp, vp]
    L.pwa_align_last_stats.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), u64p]
    L.pwa_align_batch.argtypes = batch_in + [i32p, vp, u64p, u64p, u64p, u64p]
    L.pwa_overlaps.argtypes = batch_in + [i32p, i32p]
    L.pwa_align_affine_batch.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, u64p, C.c_uint32, u32p, u32p, C.c_uint64,
                                         i32p, vp, u64p, u64p]
    L.pwa_cigar_bound.argtypes = [C.c_uint64]
    L.pwa_cigar_bound.restype = C.c_uint64
    L.pwa_mdz_bound.argtypes = [C.c_uint64]
    L.pwa_mdz_bound.restype = C.c_uint64
    L.pwa_alignment_overlap.argtypes = [vp, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64, u64p, i32p]
    L.pwa_format_alignment.argtypes = [vp, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64, u64p, vp, vp, vp, vp, i32p]
    L.pwa_fasta_read.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(vp), C.POINTER(C.c_int)]
    L.pwa_fasta_n_seq.argtypes = [vp]
    L.pwa_fasta_n_seq.restype = C.c_uint32
    L.pwa_fasta_bytes.argtypes = [vp]
    L.pwa_fasta_bytes.restype = vp
    L.pwa_fasta_offsets.argtypes = [vp]
    L.pwa_fasta_offsets.restype = u64p
    L.pwa_fasta_first_seq.argtypes = [vp]
    L.pwa_fasta_first_seq.restype = u32p
    L.pwa_fasta_free.argtypes = [vp]
    L.pwa_fasta_free.restype = None
    _lib = L
    return L


def _b(x):
    return bytes(x) if isinstance(x, (bytes, bytearray, memoryview)) else x.encode("latin-1")


def pack_sequences(seqs):
    """list of bytes -> (concatenated bytes, uint64 offsets[n+1], the list); an already packed triple passes through"""
    if (isinstance(seqs, tuple) and len(seqs) == 3 and isinstance(seqs[0], (bytes, bytearray)) and isinstance(seqs[1], C.Array)
            and getattr(seqs[1], "_type_", None) is C.c_uint64 and isinstance(seqs[2], list)):   # (a tuple of three byte strings is a list of sequences)
        return seqs
    seqs = [_b(s) for s in seqs]
    off = (C.c_uint64 * (len(seqs) + 1))()
    tot = 0
    for i, s in enumerate(seqs):
        off[i] = tot
        tot += len(s)
    off[len(seqs)] = tot
    return b"".join(seqs), off, seqs


def read_fasta(paths, n_threads=0):
    """readFasta (hw2.cpp:25-57) over one or more files -> (blob, offsets, first_seq): sequence k is
    blob[offsets[k]:offsets[k + 1]], the sequences of paths[i] are first_seq[i] .. first_seq[i + 1] - 1."""
    L = lib()
    if isinstance(paths, (str, bytes)):
        paths = [paths]
    arr = (C.c_char_p * max(len(paths), 1))(*[os.fsencode(p) for p in paths])
    h, bad = C.c_void_p(), C.c_int(-1)
    rc = L.pwa_fasta_read(arr, len(paths), n_threads, C.byref(h), C.byref(bad))
    if rc != 0:
        raise PwaError("pwa_fasta_read: %s (%s)" % (L.pwa_strerror(rc).decode(), paths[bad.value] if bad.value >= 0 else "-"))
    try:
        n = L.pwa_fasta_n_seq(h)
        off = list(L.pwa_fasta_offsets(h)[:n + 1])
        first = list(L.pwa_fasta_first_seq(h)[:len(paths) + 1])
        blob = C.string_at(L.pwa_fasta_bytes(h), off[n]) if off[n] else b""
    finally:
        L.pwa_fasta_free(h)
    return blob, off, first


def alignment_overlap(pattern, text, ops, end):
    """overlapLongestExactMatch (hw2.cpp:267-278) from the op list alone."""
    L = lib()
    pattern, text, ops = _b(pattern), _b(text), _b(ops)
    ov = C.c_int32(0)
    endc = (C.c_uint64 * 2)(end[0], end[1])
    rc = L.pwa_alignment_overlap(pattern, len(pattern), text, len(text), ops, len(ops), endc, C.byref(ov))
    if rc != 0:
        raise PwaError("pwa_alignment_overlap: %s" % L.pwa_strerror(rc).decode())
    return ov.value


def format_alignment(pattern, text, ops, end):
    """Host post-processing (hw2.cpp:59-116, 164-184, 267-278) through the C ABI."""
    L = lib()
    pattern, text, ops = _b(pattern), _b(text), _b(ops)
    n_ops = len(ops)
    ap = C.create_string_buffer(n_ops + 1)
    ar = C.create_string_buffer(n_ops + 1)
    cg = C.create_string_buffer(L.pwa_cigar_bound(n_ops))
    md = C.create_string_buffer(L.pwa_mdz_bound(n_ops))
    ov = C.c_int32(0)
    endc = (C.c_uint64 * 2)(end[0], end[1])
    rc = L.pwa_format_alignment(pattern, len(pattern), text, len(text), ops, n_ops, endc, ap, ar, cg, md, C.byref(ov))
    if rc != 0:
        raise PwaError("pwa_format_alignment: %s" % L.pwa_strerror(rc).decode())
    return dict(aligned_pattern=ap.raw[:n_ops], aligned_reference=ar.raw[:n_ops], cigar=cg.value, mdz=md.value,
                overlap=ov.value)


def upgma_newick(dist_rows, names):
    """Host UPGMA + Newick (hw4.cpp:162-228) through the C ABI."""
    L = lib()
    n = len(names)
    flat = (C.c_double * max(n * n, 1))(*[float(x) for row in dist_rows for x in row])
    arr = (C.c_char_p * max(n, 1))(*[_b(x) for x in names])
    need = C.c_uint64(0)
    L.pwa_upgma_newick(flat, arr, n, None, 0, C.byref(need))
    buf = C.create_string_buffer(need.value + 1)
    rc = L.pwa_upgma_newick(flat, arr, n, buf, need.value + 1, C.byref(need))
    if rc != 0:
        raise PwaError("pwa_upgma_newick: %s" % L.pwa_strerror(rc).decode())
    return buf.value


class Context:
    """One MI355X (HIP device ordinal)."""

    def __init__(self, device=0):
        self._L = lib()
        h = C.c_void_p()
        rc = self._L.pwa_ctx_create(device, C.byref(h))
        if rc != 0:
            raise PwaError("pwa_ctx_create(%d): %s -- the HIP path is required, there is no CPU fallback"
                           % (device, self._L.pwa_strerror(rc).decode()))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.pwa_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_score_band(self, on):
        self._check(self._L.pwa_ctx_set_score_band(self._h, 1 if on else 0), "pwa_ctx_set_score_band")

    def _check(self, rc, what):
        if rc != 0:
            raise PwaError("%s: %s (%s)" % (what, self._L.pwa_strerror(rc).decode(),
                                            self._L.pwa_last_error(self._h).decode()))

    # -- scores-only pass (hw2.cpp:328-338, score field only)
    def scores(self, mode, seqs, pair_a, pair_b, match, mismatch, gap, want_end=False):
        blob, off, seqs = pack_sequences(seqs)
        n = len(pair_a)
        pa = (C.c_uint32 * max(n, 1))(*pair_a)
        pb = (C.c_uint32 * max(n, 1))(*pair_b)
        sc = (C.c_int32 * max(n, 1))()
        ei = (C.c_uint32 * max(n, 1))() if want_end else None
        ej = (C.c_uint32 * max(n, 1))() if want_end else None
        rc = self._L.pwa_scores(self._h, MODE[mode], match, mismatch, gap, blob, off, len(seqs), pa, pb, n, sc, ei, ej)
        self._check(rc, "pwa_scores")
        if want_end:
            return list(sc[:n]), list(ei[:n]), list(ej[:n])
        return list(sc[:n])

    def batch(self, mode, seqs, pair_a, pair_b, match, mismatch, gap, want_end=False):
        return Batch(self, mode, seqs, pair_a, pair_b, match, mismatch, gap, want_end)

    # -- hw3.cpp affine_alignment score pass (hw3.cpp:23-102, all-pairs loop 232-241)
    def scores_affine(self, seqs, pair_a, pair_b, match, mismatch, gap_open, gap_extend):
        b = Batch(self, "affine", seqs, pair_a, pair_b, match, mismatch, gap_open, gap_extend=gap_extend)
        b.run()
        out = b.fetch()
        b.close()
        return out

    # -- hw4.cpp all-pairs step: NW (tie-break diag >= up >= left) + gap/mismatch column count (16-72, 146-152)
    def distances(self, seqs, pair_a, pair_b, match, mismatch, gap):
        b = Batch(self, "nwdist", seqs, pair_a, pair_b, match, mismatch, gap)
        b.run()
        out = b.fetch()
        b.close()
        return out

    def _oneshot(self, fn, name, seqs, pair_a, pair_b, *scoring):
        blob, off, seqs = pack_sequences(seqs)
        n = len(pair_a)
        pa = (C.c_uint32 * max(n, 1))(*pair_a)
        pb = (C.c_uint32 * max(n, 1))(*pair_b)
        out = (C.c_int32 * max(n, 1))()
        self._check(fn(self._h, *scoring, blob, off, len(seqs), pa, pb, n, out), name)
        return list(out[:n])

    def distances_oneshot(self, seqs, pair_a, pair_b, match, mismatch, gap):
        """pwa_distances: the one-call form (pair lists of any size: cut into arena-sized runs by the library)."""
        return self._oneshot(self._L.pwa_distances, "pwa_distances", seqs, pair_a, pair_b, match, mismatch, gap)

    def scores_affine_oneshot(self, seqs, pair_a, pair_b, match, mismatch, gap_open, gap_extend):
        """pwa_scores_affine: the one-call form."""
        return self._oneshot(self._L.pwa_scores_affine, "pwa_scores_affine", seqs, pair_a, pair_b, match, mismatch, gap_open, gap_extend)

    def batch_distances(self, seqs, pair_a, pair_b, match, mismatch, gap):
        return Batch(self, "nwdist", seqs, pair_a, pair_b, match, mismatch, gap)

    def batch_affine(self, seqs, pair_a, pair_b, match, mismatch, gap_open, gap_extend):
        return Batch(self, "affine", seqs, pair_a, pair_b, match, mismatch, gap_open, gap_extend=gap_extend)

    # -- one full alignment = one call of hw2.cpp:118 / 192
    def align(self, mode, pattern, text, match, mismatch, gap, raw=False):
        pattern, text = _b(pattern), _b(text)
        n, m = len(pattern), len(text)
        ops = C.create_string_buffer(n + m + 1)
        score = C.c_int32(0)
        n_ops = C.c_uint64(0)
        end = (C.c_uint64 * 2)()
        start = (C.c_uint64 * 2)()
        rc = self._L.pwa_align(self._h, MODE[mode], match, mismatch, gap, pattern, n, text, m, C.byref(score), ops,
                               n + m, C.byref(n_ops), end, start)
        self._check(rc, "pwa_align")
        out = dict(score=score.value, ops=ops.raw[:n_ops.value], end=(end[0], end[1]), start=(start[0], start[1]))
        if not raw:
            out.update(format_alignment(pattern, text, out["ops"], out["end"]))
        return out

    def matrices(self, mode, pattern, text, match, mismatch, gap):
        """(dp, traceback) as numpy arrays (n+1, m+1): the reference's two per-pair matrices (hw2.cpp:119-120)."""
        import numpy as np
        pattern, text = _b(pattern), _b(text)
        n, m = len(pattern), len(text)
        dp = np.zeros((n + 1, m + 1), dtype=np.int32)
        tb = np.zeros((n + 1, m + 1), dtype=np.uint8)
        rc = self._L.pwa_align_matrices(self._h, MODE[mode], match, mismatch, gap, pattern, n, text, m,
                                        dp.ctypes.data_as(C.c_void_p), tb.ctypes.data_as(C.c_void_p))
        self._check(rc, "pwa_align_matrices")
        return dp, tb

    def align_batch(self, mode, seqs, pair_a, pair_b, match, mismatch, gap, region_pad=0):
        """region_pad > 0: every pair's op region starts at a multiple of region_pad (a caller with aligned regions; the library then
        returns the op lists through its staging copy instead of one tiled copy)."""
        blob, off, seqs = pack_sequences(seqs)
        n = len(pair_a)
        pa = (C.c_uint32 * max(n, 1))(*pair_a)
        pb = (C.c_uint32 * max(n, 1))(*pair_b)
        ooff = (C.c_uint64 * max(n, 1))()
        tot = 0
        for k in range(n):
            if region_pad:
                tot = (tot + region_pad - 1) // region_pad * region_pad + (region_pad if k % 3 == 1 else 0)
            ooff[k] = tot
            tot += len(seqs[pair_a[k]]) + len(seqs[pair_b[k]])
        ops = C.create_string_buffer(tot + 1)
        sc = (C.c_int32 * max(n, 1))()
        nops = (C.c_uint64 * max(n, 1))()
        endc = (C.c_uint64 * (2 * max(n, 1)))()
        startc = (C.c_uint64 * (2 * max(n, 1)))()
        rc = self._L.pwa_align_batch(self._h, MODE[mode], match, mismatch, gap, blob, off, len(seqs), pa, pb, n, sc, ops,
                                     ooff, nops, endc, startc)
        self._check(rc, "pwa_align_batch")
        res = []
        raw = memoryview(ops)   # no per-pair copy of the whole buffer
        for k in range(n):
            o = bytes(raw[ooff[k]:ooff[k] + nops[k]])
            res.append(dict(score=sc[k], ops=o, end=(endc[2 * k], endc[2 * k + 1]),
                            start=(startc[2 * k], startc[2 * k + 1])))
        return res

    def align_batch_arrays(self, mode, packed, pair_a, pair_b, match, mismatch, gap, out=None):
        """pwa_align_batch on caller-held buffers, the way a compiled host calls it: `packed` = pack_sequences(seqs) done once,
        pair_a / pair_b numpy uint32 arrays, `out` the dict a previous call returned (its arrays are reused, nothing is
        allocated or converted per call).  Returns dict(scores, n_ops, ops, ops_off, end, start) of numpy arrays; the ops of
        pair k are ops[ops_off[k] : ops_off[k] + n_ops[k]]."""
        import numpy as np
        blob, off, seqs = packed
        n = len(pair_a)
        if out is None:
            lens = np.array([len(x) for x in seqs], dtype=np.uint64)
            cap = lens[pair_a] + lens[pair_b]
            ops_off = np.zeros(max(n, 1), dtype=np.uint64)
            if n > 1:
                ops_off[1:n] = np.cumsum(cap[:-1])
            out = dict(scores=np.zeros(max(n, 1), dtype=np.int32), n_ops=np.zeros(max(n, 1), dtype=np.uint64),
                       ops=np.zeros(int(cap.sum()) + 1, dtype=np.uint8), ops_off=ops_off,
                       end=np.zeros((max(n, 1), 2), dtype=np.uint64), start=np.zeros((max(n, 1), 2), dtype=np.uint64))
        u32p, u64p, i32p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int32)
        rc = self._L.pwa_align_batch(self._h, MODE[mode], match, mismatch, gap, blob, off, len(seqs),
                                     pair_a.ctypes.data_as(u32p), pair_b.ctypes.data_as(u32p), n, out["scores"].ctypes.data_as(i32p),
                                     out["ops"].ctypes.data_as(C.c_void_p), out["ops_off"].ctypes.data_as(u64p),
                                     out["n_ops"].ctypes.data_as(u64p), out["end"].ctypes.data_as(u64p), out["start"].ctypes.data_as(u64p))
        self._check(rc, "pwa_align_batch")
        return out

    def align_affine_batch(self, seqs, pair_a, pair_b, match, mismatch, gap_open, gap_extend):
        """hw3.cpp:23-135 for a pair list -> [dict(score, ops)], ops in traceback order ('M' / 'D' / 'I')."""
        blob, off, seqs = pack_sequences(seqs)
        n = len(pair_a)
        pa = (C.c_uint32 * max(n, 1))(*pair_a)
        pb = (C.c_uint32 * max(n, 1))(*pair_b)
        ooff = (C.c_uint64 * max(n, 1))()
        tot = 0
        for k in range(n):
            ooff[k] = tot
            tot += len(seqs[pair_a[k]]) + len(seqs[pair_b[k]])
        ops = C.create_string_buffer(tot + 1)
        sc = (C.c_int32 * max(n, 1))()
        nops = (C.c_uint64 * max(n, 1))()
        rc = self._L.pwa_align_affine_batch(self._h, match, mismatch, gap_open, gap_extend, blob, off, len(seqs), pa, pb, n, sc,
                                            ops, ooff, nops)
        self._check(rc, "pwa_align_affine_batch")
        raw = memoryview(ops)
        return [dict(score=sc[k], ops=bytes(raw[ooff[k]:ooff[k] + nops[k]])) for k in range(n)]

    def overlaps(self, mode, seqs, pair_a, pair_b, match, mismatch, gap):
        """(scores, overlaps) of full alignments without their op lists: the -g selection inputs (hw2.cpp:342-350)."""
        blob, off, seqs = pack_sequences(seqs)
        n = len(pair_a)
        pa = (C.c_uint32 * max(n, 1))(*pair_a)
        pb = (C.c_uint32 * max(n, 1))(*pair_b)
        sc = (C.c_int32 * max(n, 1))()
        ov = (C.c_int32 * max(n, 1))()
        rc = self._L.pwa_overlaps(self._h, MODE[mode], match, mismatch, gap, blob, off, len(seqs), pa, pb, n, sc, ov)
        self._check(rc, "pwa_overlaps")
        return list(sc[:n]), list(ov[:n])

    def align_stats(self):
        f, t, b = C.c_float(0), C.c_float(0), C.c_uint64(0)
        self._L.pwa_align_last_stats(self._h, C.byref(f), C.byref(t), C.byref(b))
        return dict(fill_ms=f.value, traceback_ms=t.value, band_bytes=b.value)


class Batch:
    """Prepared scores-only batch: inputs resident in HBM, run() only enqueues kernels."""

    def __init__(self, ctx, mode, seqs, pair_a, pair_b, match, mismatch, gap, want_end=False, gap_extend=0):
        self._ctx, self._L = ctx, ctx._L
        blob, off, seqs = pack_sequences(seqs)
        self.n_pairs = len(pair_a)
        n = self.n_pairs
        if hasattr(pair_a, "ctypes"):   # numpy uint32 arrays
            pa = pair_a.ctypes.data_as(C.POINTER(C.c_uint32))
            pb = pair_b.ctypes.data_as(C.POINTER(C.c_uint32))
            self._keep = (pair_a, pair_b)
        else:
            pa = (C.c_uint32 * max(n, 1))(*pair_a)
            pb = (C.c_uint32 * max(n, 1))(*pair_b)
        h = C.c_void_p()
        if mode == "nwdist":
            rc = self._L.pwa_nwdist_batch_create(ctx._h, match, mismatch, gap, blob, off, len(seqs), pa, pb, n, C.byref(h))
        elif mode == "affine":
            rc = self._L.pwa_affine_batch_create(ctx._h, match, mismatch, gap, gap_extend, blob, off, len(seqs), pa, pb, n,
                                                 C.byref(h))
        else:
            rc = self._L.pwa_batch_create(ctx._h, MODE[mode], match, mismatch, gap, blob, off, len(seqs), pa, pb, n,
                                          1 if want_end else 0, C.byref(h))
        ctx._check(rc, "pwa_batch_create")
        self._h = h
        self.want_end = want_end

    def run(self, stream=None):
        self._ctx._check(self._L.pwa_batch_run(self._h, stream), "pwa_batch_run")

    def d_scores(self):
        return self._L.pwa_batch_d_scores(self._h)

    def set_d_scores(self, device_ptr):
        """Redirect the score vector to caller-owned device memory (e.g. tensor.data_ptr())."""
        self._ctx._check(self._L.pwa_batch_set_d_scores(self._h, device_ptr), "pwa_batch_set_d_scores")

    def last_ms(self):
        ms = C.c_float(0)
        self._ctx._check(self._L.pwa_batch_last_ms(self._h, C.byref(ms)), "pwa_batch_last_ms")
        return ms.value

    def run_times(self, cap=64):
        arr = (C.c_float * cap)()
        n = C.c_int(0)
        self._ctx._check(self._L.pwa_batch_run_times(self._h, arr, cap, C.byref(n)), "pwa_batch_run_times")
        return list(arr[:n.value])

    def info(self):
        cells, padded, nt = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        name = C.c_char_p()
        self._L.pwa_batch_info(self._h, C.byref(cells), C.byref(padded), C.byref(nt), C.byref(name))
        return dict(cells=cells.value, padded_cells=padded.value, n_tasks=nt.value, kernel=name.value.decode())

    def fetch(self, numpy_out=False):
        n = self.n_pairs
        sc = (C.c_int32 * max(n, 1))()
        ei = (C.c_uint32 * max(n, 1))() if self.want_end else None
        ej = (C.c_uint32 * max(n, 1))() if self.want_end else None
        self._ctx._check(self._L.pwa_batch_fetch(self._h, sc, ei, ej), "pwa_batch_fetch")
        if numpy_out:
            import numpy as np
            s = np.ctypeslib.as_array(sc)[:n].copy()
            return (s, np.ctypeslib.as_array(ei)[:n].copy(), np.ctypeslib.as_array(ej)[:n].copy()) if self.want_end else s
        if self.want_end:
            return list(sc[:n]), list(ei[:n]), list(ej[:n])
        return list(sc[:n])

    def fetch_into(self, scores):
        """pwa_batch_fetch straight into a caller-owned C-contiguous numpy int32 array of n_pairs elements."""
        assert scores.dtype.str in ("<i4", "=i4") and scores.flags["C_CONTIGUOUS"] and scores.size == self.n_pairs
        self._ctx._check(self._L.pwa_batch_fetch(self._h, scores.ctypes.data_as(C.POINTER(C.c_int32)), None, None), "pwa_batch_fetch")

    def close(self):
        if getattr(self, "_h", None):
            self._L.pwa_batch_destroy(self._h)
            self._h = None

    __del__ = close
