/*
 * hw2_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the pairwise-alignment path of the reference
 * program Local_Global_Alignment/hw2.cpp.  It exists to CHECK the HIP path; it is
 * never part of the product.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle.py)
 * against the reference's own golden outputs global.txt / local.txt, against the
 * known answers of SURVEY.md section 8(a)/(d), and -- in the dev container --
 * differentially against the unmodified hw2.cpp compiled into oracle/_ref/.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference/Local_Global_Alignment/).
 */
#ifndef HW2_ORACLE_H
#define HW2_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors struct AlignmentResult, hw2.cpp:17-23, plus the raw traceback walk. */
typedef struct orc_result {
    int32_t score;
    char *aligned_pattern;   /* NUL-terminated, forward order (after the reverse at 183/258) */
    char *aligned_reference; /* idem (184/259) */
    char *cigar;             /* prepareCigarString, hw2.cpp:59-78 */
    char *mdz;               /* prepareMDZString,   hw2.cpp:80-116 */
    char *ops;               /* 'M'/'D'/'I' in TRACEBACK order (end -> start), n_ops bytes + NUL */
    size_t n_ops;
    size_t end_i, end_j;     /* cell the walk starts from: NW (n,m); SW first row-major argmax */
    size_t start_i, start_j; /* cell the walk stops at */
} orc_result;

/* hw2.cpp:118-190 globalAlignmentNeedlemanWunsch, full (n+1)(m+1) int + char matrices. */
orc_result *orc_nw(const char *p, size_t n, const char *t, size_t m, int match, int mismatch, int gap);
/* hw2.cpp:192-265 localAlignmentSmithWaterman, full matrices. */
orc_result *orc_sw(const char *p, size_t n, const char *t, size_t m, int match, int mismatch, int gap);
void orc_free(orc_result *r);

/* The reference's two matrices themselves (hw2.cpp:119-156 / 193-231): dp int32 and traceback char,
 * (n+1) x (m+1) row-major, written into caller buffers.  mode 0 = NW, 1 = SW. */
void orc_matrices(int mode, const char *p, size_t n, const char *t, size_t m, int match, int mismatch, int gap,
                  int32_t *dp, char *tb);

/* Same results as orc_nw / orc_sw but with two rolling int rows + a 1 B/cell code
 * matrix (for sizes where 5 B/cell does not fit); differential-tested against the
 * full-matrix forms. */
orc_result *orc_nw_compact(const char *p, size_t n, const char *t, size_t m, int match, int mismatch, int gap);
orc_result *orc_sw_compact(const char *p, size_t n, const char *t, size_t m, int match, int mismatch, int gap);

/* Score-only forms (O(m) memory): recurrences of hw2.cpp:138-156 and 205-231.
 * orc_sw_score also returns the first row-major argmax (hw2.cpp:225-229). */
int32_t orc_nw_score(const char *p, size_t n, const char *t, size_t m, int match, int mismatch, int gap);
int32_t orc_sw_score(const char *p, size_t n, const char *t, size_t m, int match, int mismatch, int gap,
                     size_t *end_i, size_t *end_j);

/* hw2.cpp:267-278 overlapLongestExactMatch. */
int orc_overlap(const char *aligned_pattern, const char *aligned_reference, size_t len);

/* hw2.cpp:59-78 / 80-116 on an op list in traceback order. Caller frees. */
char *orc_cigar(const char *ops_tb_order, size_t n_ops);
char *orc_mdz(const char *aligned_pattern, const char *aligned_reference, const char *ops_tb_order, size_t n_ops);

/* hw2.cpp:25-57 readFasta. Returns 0, or -1 if the file cannot be opened
 * (the reference prints "Error: Cannot open file X" and exit(1)s: 28-31). */
typedef struct orc_fasta {
    size_t count;
    char **seq;
    size_t *len;
} orc_fasta;
int orc_read_fasta(const char *path, orc_fasta *out);
void orc_free_fasta(orc_fasta *f);

/* hw2.cpp:280-403 main(): same argv handling, stderr text, exit codes, output bytes. */
int orc_hw2_main(int argc, char **argv);

/* The synthetic generator of SURVEY.md section 8(d) (splitmix64, counter based). */
void orc_gen(uint64_t seed, uint64_t stream, uint64_t id, size_t len, char *out);

#ifdef __cplusplus
}
#endif
#endif
