/*
 * hw4_oracle.h -- TEST INFRASTRUCTURE ONLY (same rules as hw2_oracle.h).
 * CPU restatement (plain C) of /root/reference/hw4/hw4.cpp: needleman_wunsch (16-72, tie-break
 * diag >= up >= left), the traceback-derived distance (146-152), the FASTA parser (100-136), UPGMA
 * (162-226) and the Newick output (228-237).
 * Parity status: PINNED against the unmodified hw4.cpp compiled into oracle/_ref/ (fixtures in
 * tests/golden/hw4_*.json, the reference's own input.fasta -> tree.txt).
 */
#ifndef HW4_ORACLE_H
#define HW4_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* distance = alignment columns that hold a gap or a mismatch (hw4.cpp:146-152); *score = dp[n][m]. */
int32_t orc4_nw_distance(const char *s1, size_t n, const char *s2, size_t m, int match, int mismatch, int gap, int32_t *score);
/* UPGMA over a dense symmetric distance matrix (row-major doubles) -> Newick string incl. ":0.0;" (malloc'ed). */
char *orc4_upgma(const double *dist, const char *const *names, size_t n);
/* whole program; same argv, stderr texts, exit codes and output bytes as hw4.cpp:74-240 */
int orc4_main(int argc, char **argv);
#ifdef __cplusplus
}
#endif
#endif
