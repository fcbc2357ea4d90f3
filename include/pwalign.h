/*
 * pwalign.h -- C ABI of the MI355X-native pairwise-alignment engine (libpwalign.so).
 *
 * Drop-in boundary for the DP hot path of the reference program
 *   /root/reference/Local_Global_Alignment/hw2.cpp
 * The reference has no FFI; its only callable surface for this path is the two C++ functions
 *   AlignmentResult* globalAlignmentNeedlemanWunsch(const string&, const string&, int, int, int)  hw2.cpp:118
 *   AlignmentResult* localAlignmentSmithWaterman  (const string&, const string&, int, int, int)  hw2.cpp:192
 * called once per pair from the loop at hw2.cpp:328-338.  Each entry point below names the
 * reference lines it replaces.  INTEGRATION.md shows the stub a maintainer of hw2.cpp would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every buffer is caller-owned HOST memory unless the
 *     parameter name starts with d_ (then it is a DEVICE pointer on the context's GPU);
 *   - sequences are RAW BYTES compared for equality (hw2.cpp:142, 208): any alphabet, case-sensitive;
 *   - scores are int32 with the reference's recurrences and tie-breaks; results are bit-identical
 *     to hw2.cpp wherever hw2.cpp itself does not overflow int -- for ANY match / mismatch / gap: the fills with a
 *     traceback band keep H*4+priority keys while max|score| * (n + m + 2) <= 2^28 and switch to a plain int32
 *     compare-and-select form beyond that (slower, never refused);
 *   - every function returns PWA_OK (0) or a negative PWA_E_* code; nothing throws, exits or
 *     prints across the ABI; pwa_last_error() gives the text of the last failure on a context;
 *   - a context is bound to ONE GPU and is not thread-safe: one context per host thread.
 *   - there is NO CPU fallback: without a usable GPU pwa_ctx_create fails with PWA_E_NODEVICE.
 *
 * Engines (chosen by the library, never by the caller; all exact, so the choice changes no result):
 *   strip engine       scores only: lane = pair, register strips (batch_scores.hip.h);
 *   stripe engine      one wave per 64 RL rows of a pair, anti-diagonal front, stripes pipelined (pair_fill.hip.h): traceback fills
 *                      of long patterns, scores with end cells, and -- by estimated cost -- the scores of pairs that would leave the
 *                      strip engine's waves under-filled (a few long pairs);
 *   mini-stripe engine 16 lanes per pair, four pairs per wave (mini_fill.hip.h): traceback fills of patterns of up to 256 rows, and --
 *                      without a band -- the scores of such pairs when a scores pass routes them off the strips or asks for end cells.
 *   pwa_align_batch / pwa_overlaps pick the band geometry pair by pair; pwa_scores / pwa_batch_create split a list between the three
 *   engines by estimated cost.
 *
 * Environment switches (tests and diagnostics only).  They are read ONCE, by pwa_ctx_create, into the context; no other entry point
 * consults the environment, so a process that wants another setting creates another context.  (pwa_fasta_read, which has no context,
 * reads PWA_FASTA_MIN_CHUNK -- bytes per parser chunk -- on every call.)
 *   PWA_DEBUG, PWA_PROBE          host-side phase times / nop-kernel probes on stderr
 *   PWA_SCORES_ROUTE=0|1          scores passes: 0 every pair on the strip engine, 1 every pair on the stripe engine (default: by cost)
 *   PWA_TB_ENGINE=0|2             traceback fills and scores off the strips: 0 the stripe engine's plain forms only, 2 mini-stripe kernels
 *                                 wherever they exist (default: by the list -- patterns of <= 256 rows, and of <= 1024 rows in batches)
 *   PWA_NO_PIPELINE, PWA_PIPE_RUNS=N  one-shot score calls: runs strictly one after the other / a list that fits one arena cut into N runs
 *   PWA_ARENA_LIMIT, PWA_LANE_ROWS_LIMIT   bytes per run of the one-shot calls / per-lane text rows per batch (force the multi-run paths)
 *   PWA_RANGE_BYTES               band + op bytes per range of pwa_align_batch / pwa_overlaps (forces several ranges on a small list)
 *   PWA_NO_PAIR_TABLE, PWA_NO_KEYED_TB, PWA_NO_GAP_SHIFT, PWA_NO_TILED_OPS, PWA_NO_PACKED_DIST, PWA_PAIRED, PWA_FORCE_LANES,
 *   PWA_FORCE_R, PWA_FORCE_MODE, PWA_FORCE_RL, PWA_FORCE_W, PWA_WG_PER_CU, PWA_MINI_PER_CU, PWA_NO_LDS_PAD, PWA_STRIP_WG1, PWA_STAMPS,
 *   PWA_TRACE_STRIPE
 *                                 select one of several equivalent kernel forms / geometries, or record time stamps
 */
#ifndef PWALIGN_H
#define PWALIGN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PWA_MODE_NW 0 /* global, hw2.cpp:118-190 (tie-break diag >= left >= up, 145-153)      */
#define PWA_MODE_SW 1 /* local,  hw2.cpp:192-265 (tie-break zero > diag > up > left, 214-222) */

#define PWA_OK 0
#define PWA_E_INVALID (-1)     /* bad argument (null pointer, unknown mode, index out of range) */
#define PWA_E_NODEVICE (-2)    /* no usable gfx950 device / HIP runtime                         */
#define PWA_E_HIP (-3)         /* a HIP call failed (text in pwa_last_error)                     */
#define PWA_E_NOMEM (-4)       /* host or device allocation failed                              */
#define PWA_E_CAPACITY (-5)    /* caller buffer too small / problem exceeds an index width       */
#define PWA_E_IO (-6)          /* a file cannot be opened or read (pwa_fasta_read)                */

typedef struct pwa_ctx pwa_ctx;
typedef struct pwa_batch pwa_batch;

/* Library / build identification: "pwalign <ver> gfx950". */
const char *pwa_version(void);
const char *pwa_strerror(int code);
/* Test hook, no GPU involved: runs the host scheduler's sorting helpers (stable counting sort, its multi-threaded form, the length
 * sort, the radix sort) on pseudo-random lists against std::stable_sort.  Returns 0, or the number of the first check that failed. */
int pwa_selftest_host(uint32_t seed);

/* Context = one GPU (HIP device ordinal) + its streams and workspaces. */
int pwa_ctx_create(int device, pwa_ctx **out);
void pwa_ctx_destroy(pwa_ctx *ctx);
const char *pwa_last_error(const pwa_ctx *ctx);
/* When on, pwa_align / pwa_align_batch also materialise the int32 SCORE band in HBM next to the
 * traceback band (4 + 1 B/cell: the reference's own footprint, hw2.cpp:119-120 / 193-194), written as
 * one coalesced 64*RL*4-byte wave store per anti-diagonal step.  The walk does not need it; it exists
 * for inspection (pwa_align_matrices returns it) and for the 5 B/cell roofline accounting. */
int pwa_ctx_set_score_band(pwa_ctx *ctx, int on);

/*
 * Scores of many pairs (the scores-only pass over hw2.cpp's pair loop 328-338; for -l the
 * reference selects the best pair from `result->score` alone, 352-356).
 *
 *   seq_bytes/seq_off : n_seq sequences, sequence s = seq_bytes[seq_off[s] .. seq_off[s+1])
 *   pair_a/pair_b     : pair k aligns pattern = sequence pair_a[k] (rows, hw2 "patterns")
 *                       against reference/text = sequence pair_b[k] (columns, hw2 "references")
 *   score_out[k]      : NW: dp[n][m] (hw2.cpp:186);  SW: max cell (hw2.cpp:225-229)
 *   end_i_out/end_j_out (each may be NULL): the cell the reference's traceback starts from --
 *                       NW (n, m); SW the FIRST maximum in row-major order, (0,0) if all zero.
 * Pair lists of any size: a batch object addresses its sequence arena with 32-bit offsets (4 GiB of distinct sequences),
 * so this call (like pwa_distances and pwa_scores_affine) cuts the list into runs of consecutive pairs whose sequences
 * fit one arena and PIPELINES them: run k + 1 is scheduled, coded and uploaded while the kernels of run k execute.  Remaining limits: a sequence < 2^31 - 64 symbols; pattern +
 * reference of ONE pair < 4 GiB; < 2^32 - 1 pairs per call.
 */
int pwa_scores(pwa_ctx *ctx, int mode, int match, int mismatch, int gap, const uint8_t *seq_bytes,
               const uint64_t *seq_off, uint32_t n_seq, const uint32_t *pair_a, const uint32_t *pair_b,
               uint64_t n_pairs, int32_t *score_out, uint32_t *end_i_out, uint32_t *end_j_out);

/*
 * The same pass split into prepare / run / fetch so that a caller can keep inputs resident in
 * HBM, time the kernels alone, and hand the device-side score vector to a collective
 * (RCCL all-gather over xGMI) without a host round trip.
 *   pwa_batch_create  uploads the sequences, builds the wave-task list (host), allocates outputs; PWA_E_CAPACITY when the
 *                     sequences the pair list uses exceed one 4 GiB arena (split the list, or call pwa_scores);
 *   pwa_batch_run     enqueues the kernels on `stream` (a hipStream_t, NULL = the context's own
 *                     stream) -- asynchronous, no host synchronisation, graph-capturable;
 *   pwa_batch_d_scores  device pointer to int32[n_pairs] in pair order (valid until destroy),
 *                     complete once the run has finished on its stream;
 *   pwa_batch_fetch   synchronises the stream and copies scores (and end cells, if requested at
 *                     create time) to host buffers.
 */
int pwa_batch_create(pwa_ctx *ctx, int mode, int match, int mismatch, int gap, const uint8_t *seq_bytes,
                     const uint64_t *seq_off, uint32_t n_seq, const uint32_t *pair_a, const uint32_t *pair_b,
                     uint64_t n_pairs, int want_end_cells, pwa_batch **out);
int pwa_batch_run(pwa_batch *b, void *stream);
int32_t *pwa_batch_d_scores(pwa_batch *b);
/* Redirect the score vector to caller-owned DEVICE memory (int32[n_pairs], e.g. a torch tensor's
 * data_ptr) so that it can be handed to a collective without aliasing library memory. */
int pwa_batch_set_d_scores(pwa_batch *b, int32_t *d_scores);
int pwa_batch_fetch(pwa_batch *b, int32_t *score_out, uint32_t *end_i_out, uint32_t *end_j_out);
/* Facts about the prepared batch for reporting: cells = sum n*m; padded_cells = cells the kernels
 * actually evaluate (register-tile padding); kernel_name = the dominant kernel instantiation. */
int pwa_batch_info(const pwa_batch *b, uint64_t *cells, uint64_t *padded_cells, uint64_t *n_tasks,
                   const char **kernel_name);
/* Device time of the most recent pwa_batch_run in ms (HIP events on the run's stream); the call
 * synchronises the run. */
int pwa_batch_last_ms(pwa_batch *b, float *ms);
/* Device times (ms, oldest first) of up to `cap` most recent runs (at most 64 are kept): each is
 * bracketed by HIP events recorded on the stream the kernels were enqueued on. */
int pwa_batch_run_times(pwa_batch *b, float *ms_out, int cap, int *n_out);
void pwa_batch_destroy(pwa_batch *b);

/*
 * Affine-gap global alignment SCORES of many pairs -- the score pass of the sibling program
 *   /root/reference/Multiple_Sequence_Alignment/hw3.cpp
 * i.e. `affine_alignment(Si, Sj, M, Mm, Go, Ge, &score)` (hw3.cpp:23-102) as called by the all-pairs
 * loop of the center-star MSA (hw3.cpp:232-241).  The recurrence is hw3's own three-matrix form with
 * its quirks (boundary gap of length L costs Go + Ge(L-1), interior gap Go + Ge*L; F never reads E and
 * E never reads F; result = max(V, F, E)[n][m]); it is NOT interchangeable with hw2's linear-gap NW.
 * pair (a, b): string1 = sequence a (rows), string2 = sequence b (columns).
 * The returned batch object works with pwa_batch_run / _d_scores / _set_d_scores / _fetch / _info /
 * _run_times / _destroy exactly like a linear-gap batch (no end cells).
 */
int pwa_affine_batch_create(pwa_ctx *ctx, int match, int mismatch, int gap_open, int gap_extend, const uint8_t *seq_bytes,
                            const uint64_t *seq_off, uint32_t n_seq, const uint32_t *pair_a, const uint32_t *pair_b,
                            uint64_t n_pairs, pwa_batch **out);
int pwa_scores_affine(pwa_ctx *ctx, int match, int mismatch, int gap_open, int gap_extend, const uint8_t *seq_bytes,
                      const uint64_t *seq_off, uint32_t n_seq, const uint32_t *pair_a, const uint32_t *pair_b,
                      uint64_t n_pairs, int32_t *score_out);

/*
 * The alignments hw3.cpp builds against the center of the star (hw3.cpp:261-283): affine_alignment(string1, string2,
 * ..., &alignedString1, &alignedString2), i.e. hw3.cpp:23-135 with its three trace matrices, its tie-breaks (V,
 * then F if strictly greater, then E if strictly greater; a gap is extended only if that is strictly better than
 * opening one) and its walk.  pair (a, b): string1 = sequence a, string2 = sequence b.  Pairs that share string1
 * (the center) are processed 64 to a wavefront.
 *   score_out[k] : max(V, F, E)[n][m] as pwa_scores_affine
 *   ops          : one byte per alignment column in TRACEBACK order (end -> start), pair k at ops[ops_off[k] ..):
 *                  'M' both symbols, 'D' string1 symbol against '-', 'I' '-' against string2 symbol; the caller
 *                  leaves room for n_k + m_k bytes per pair;   n_ops[k] : columns of pair k
 */
int pwa_align_affine_batch(pwa_ctx *ctx, int match, int mismatch, int gap_open, int gap_extend, const uint8_t *seq_bytes,
                           const uint64_t *seq_off, uint32_t n_seq, const uint32_t *pair_a, const uint32_t *pair_b,
                           uint64_t n_pairs, int32_t *score_out, uint8_t *ops, const uint64_t *ops_off, uint64_t *n_ops);

/*
 * The all-pairs step of the sibling program /root/reference/hw4/hw4.cpp (138-159): per pair a
 * Needleman-Wunsch alignment with hw4's tie-break (diag >= up >= left, hw4.cpp:36-47 -- not hw2's) and
 * the number of alignment columns that hold a gap or a mismatch (146-152).  dist_out[k] is that count
 * for pair k (pair_a = sequence1 = rows, pair_b = sequence2 = columns).  No traceback is stored: the
 * kernel carries the distance of the chosen path through the DP.  The batch object behaves like any
 * other (run / d_scores / fetch / info / destroy; no end cells).
 */
int pwa_nwdist_batch_create(pwa_ctx *ctx, int match, int mismatch, int gap, const uint8_t *seq_bytes,
                            const uint64_t *seq_off, uint32_t n_seq, const uint32_t *pair_a, const uint32_t *pair_b,
                            uint64_t n_pairs, pwa_batch **out);
int pwa_distances(pwa_ctx *ctx, int match, int mismatch, int gap, const uint8_t *seq_bytes, const uint64_t *seq_off,
                  uint32_t n_seq, const uint32_t *pair_a, const uint32_t *pair_b, uint64_t n_pairs, int32_t *dist_out);
/* Host-side UPGMA + Newick of hw4.cpp:162-228 over a dense symmetric n x n matrix of doubles (no GPU
 * work).  Writes the NUL-terminated tree "(...):0.0;" into out; *needed receives the size required
 * (PWA_E_CAPACITY when cap is too small; call with cap = 0 to size the buffer). */
int pwa_upgma_newick(const double *dist, const char *const *names, uint32_t n, char *out, uint64_t cap, uint64_t *needed);

/*
 * Full alignment of ONE pair: matrix fill with the traceback band in HBM + traceback walk on
 * the device.  Replaces one call of hw2.cpp:118 / hw2.cpp:192 up to (not including) the string
 * post-processing prepareCigarString / prepareMDZString (59-116), which stays on the host.
 *
 *   ops      : receives the walk's op bytes 'M' (diagonal), 'D' (up: pattern char vs '-'),
 *              'I' (left: '-' vs text char) in TRACEBACK order, i.e. exactly the contents of
 *              the reference's `tracebacks` vector (hw2.cpp:161, 237) before any reversal;
 *   ops_cap  : capacity of ops; n + m always suffices (PWA_E_CAPACITY otherwise);
 *   end_cell : {i, j} the walk starts from (NW: n, m; SW: first row-major maximum);
 *   start_cell: {i, j} where it stops (NW: 0,0).   Either may be NULL.
 */
int pwa_align(pwa_ctx *ctx, int mode, int match, int mismatch, int gap, const uint8_t *pattern, uint64_t n,
              const uint8_t *text, uint64_t m, int32_t *score, uint8_t *ops, uint64_t ops_cap, uint64_t *n_ops,
              uint64_t end_cell[2], uint64_t start_cell[2]);

/*
 * The two matrices the reference keeps per pair, in the reference's own row-major form (for
 * inspection and whole-matrix parity tests; sizes are the caller's problem: 5 B per cell):
 *   dp_out : int32 (n+1) x (m+1) = `dp`        (hw2.cpp:119 / 193), or NULL
 *   tb_out : char  (n+1) x (m+1) = `traceback` (hw2.cpp:120 / 194): ' ', 'd', 'u', 'l', '0', or NULL
 * On the device both are written as skewed bands (int32 score band + 1 B/cell traceback band, one
 * coalesced wave store per anti-diagonal step); the host un-skews them.
 */
int pwa_align_matrices(pwa_ctx *ctx, int mode, int match, int mismatch, int gap, const uint8_t *pattern, uint64_t n,
                       const uint8_t *text, uint64_t m, int32_t *dp_out, char *tb_out);

/* Device time in ms of the fill kernel(s) / traceback kernel of the last pwa_align on ctx, and
 * the bytes of traceback band it wrote to HBM (for roofline accounting). */
int pwa_align_last_stats(const pwa_ctx *ctx, float *fill_ms, float *traceback_ms, uint64_t *band_bytes);

/*
 * Full alignment of many pairs (the -g path needs every pair's alignment: hw2.cpp:344).
 * ops of pair k are written at ops[ops_off[k] .. ops_off[k] + n_ops[k]); the caller sizes
 * ops_off so that pair k has room for n_k + m_k bytes.  When the regions follow one another without a gap
 * (ops_off[k + 1] == ops_off[k] + n_k + m_k) the device op buffer mirrors the caller's and every chunk of pairs comes
 * back with one copy straight into `ops`; any other layout works through a staging copy.  No sequence-arena limit here
 * (64-bit device pointers); the traceback bands are processed in ranges of consecutive pairs that fit the free HBM, and inside a
 * range every pair runs with the band geometry its own pattern length asks for.  Bytes of a pair's region beyond n_ops[k] are
 * undefined after the call.
 */
int pwa_align_batch(pwa_ctx *ctx, int mode, int match, int mismatch, int gap, const uint8_t *seq_bytes,
                    const uint64_t *seq_off, uint32_t n_seq, const uint32_t *pair_a, const uint32_t *pair_b,
                    uint64_t n_pairs, int32_t *score_out, uint8_t *ops, const uint64_t *ops_off, uint64_t *n_ops,
                    uint64_t *end_cells /* 2*n_pairs or NULL */, uint64_t *start_cells /* 2*n_pairs or NULL */);

/*
 * The -g selection without the op lists: hw2.cpp:342-350 keeps, of every pair's global alignment, only
 * overlapLongestExactMatch(alignedPattern, alignedReference) (hw2.cpp:267-278) and the score.  Same fill and
 * traceback band as pwa_align_batch; the device walk looks at the symbols under each run of diagonal moves
 * and returns the longest run of equal, gap-free columns -- no op list is written or copied back.  The
 * caller then asks pwa_align for the ONE winning pair (first strictly larger overlap, hw2.cpp:346).
 *   score_out[k]   : as pwa_align_batch;   overlap_out[k] : as pwa_alignment_overlap on that pair's walk
 */
int pwa_overlaps(pwa_ctx *ctx, int mode, int match, int mismatch, int gap, const uint8_t *seq_bytes,
                 const uint64_t *seq_off, uint32_t n_seq, const uint32_t *pair_a, const uint32_t *pair_b, uint64_t n_pairs,
                 int32_t *score_out, int32_t *overlap_out);

/*
 * FASTA ingest (no GPU work): readFasta (hw2.cpp:25-57) on one or more files, straight into the layout the
 * entries above consume -- ONE byte blob and n_seq + 1 offsets -- parsed by n_threads threads (<= 0: one per
 * host core, at most 16) over the mmap'ed file.  Semantics are the reference's, byte for byte: header text is
 * dropped, lines are joined after stripping trailing '\r' / whitespace, blank lines are skipped, records with an
 * empty body are dropped, bytes ahead of the first header form a record.  Sequences of paths[i] are
 * first_seq[i] .. first_seq[i + 1] - 1.  PWA_E_IO when a file cannot be opened (the reference prints
 * "Error: Cannot open file <name>" and exits 1, hw2.cpp:28-31): *failed_path is its index.
 */
typedef struct pwa_fasta pwa_fasta;
int pwa_fasta_read(const char *const *paths, int n_paths, int n_threads, pwa_fasta **out, int *failed_path /* or NULL */);
uint32_t pwa_fasta_n_seq(const pwa_fasta *f);
const uint8_t *pwa_fasta_bytes(const pwa_fasta *f);
const uint64_t *pwa_fasta_offsets(const pwa_fasta *f);   /* n_seq + 1 */
const uint32_t *pwa_fasta_first_seq(const pwa_fasta *f); /* n_paths + 1 */
void pwa_fasta_free(pwa_fasta *f);

/*
 * Host-side post-processing of one alignment (no GPU work): everything hw2.cpp derives from
 * the walk -- the gapped strings (hw2.cpp:164-184 / 240-259), prepareCigarString (59-78),
 * prepareMDZString (80-116) and overlapLongestExactMatch (267-278) -- so that the five fields
 * of the reference's AlignmentResult (hw2.cpp:17-23) can be rebuilt from pwa_align's output.
 *   ops / n_ops / end_cell : as returned by pwa_align (ops in traceback order)
 *   aligned_pattern, aligned_reference : n_ops + 1 bytes each (NUL-terminated)
 *   cigar : pwa_cigar_bound(n_ops) bytes;  mdz : pwa_mdz_bound(n_ops) bytes
 * Note the reference's conventions are kept verbatim: 'D' = pattern char against '-',
 * 'I' = '-' against text char, MD:Z mismatches print the REFERENCE character and deletions
 * print the PATTERN characters.
 */
/* overlapLongestExactMatch (hw2.cpp:267-278) straight from the op list, without building the strings:
 * what the -g selection loop (hw2.cpp:342-350) needs for every pair. */
int pwa_alignment_overlap(const uint8_t *pattern, uint64_t n, const uint8_t *text, uint64_t m, const uint8_t *ops,
                          uint64_t n_ops, const uint64_t end_cell[2], int32_t *overlap);
uint64_t pwa_cigar_bound(uint64_t n_ops);
uint64_t pwa_mdz_bound(uint64_t n_ops);
int pwa_format_alignment(const uint8_t *pattern, uint64_t n, const uint8_t *text, uint64_t m, const uint8_t *ops,
                         uint64_t n_ops, const uint64_t end_cell[2], char *aligned_pattern, char *aligned_reference,
                         char *cigar, char *mdz, int32_t *overlap);

#ifdef __cplusplus
}
#endif
#endif /* PWALIGN_H */
