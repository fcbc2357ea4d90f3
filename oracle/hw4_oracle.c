/* hw4_oracle.c -- TEST INFRASTRUCTURE ONLY (see hw4_oracle.h). */
#include "hw4_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline int32_t add4(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static char *dup4(const char *s) {
    size_t n = strlen(s) + 1;
    char *p = (char *)malloc(n);
    memcpy(p, s, n);
    return p;
}

int32_t orc4_nw_distance(const char *s1, size_t n, const char *s2, size_t m, int match, int mismatch, int gap, int32_t *score) {
    const size_t W = m + 1;
    int32_t *dp = (int32_t *)calloc((n + 1) * W, sizeof(int32_t));   /* 18 */
    char *tb = (char *)calloc((n + 1) * W, 1);                       /* 19 */
    for (size_t i = 1; i <= n; ++i) {                                /* 21-24 */
        dp[i * W] = add4(dp[(i - 1) * W], gap);
        tb[i * W] = 'U';
    }
    for (size_t j = 1; j <= m; ++j) {                                /* 25-28 */
        dp[j] = add4(dp[j - 1], gap);
        tb[j] = 'L';
    }
    for (size_t i = 1; i <= n; ++i)
        for (size_t j = 1; j <= m; ++j) {                            /* 30-48 */
            int32_t up = add4(dp[(i - 1) * W + j], gap), left = add4(dp[i * W + j - 1], gap);
            int32_t v = add4(dp[(i - 1) * W + j - 1], s1[i - 1] == s2[j - 1] ? match : mismatch);
            char c = 'D';
            if (up > v) { v = up; c = 'U'; }                         /* 40-43: up before left */
            if (left > v) { v = left; c = 'L'; }                     /* 44-47 */
            dp[i * W + j] = v;
            tb[i * W + j] = c;
        }
    size_t i = n, j = m;
    int32_t dist = 0;
    while (i > 0 || j > 0) {                                         /* 52-66 + 146-152 */
        if (i > 0 && j > 0 && tb[i * W + j] == 'D') {
            if (s1[i - 1] != s2[j - 1]) ++dist;
            --i; --j;
        } else if (i > 0 && tb[i * W + j] == 'U') {
            ++dist; --i;
        } else {
            ++dist; --j;
        }
    }
    if (score) *score = dp[n * W + m];
    free(dp);
    free(tb);
    return dist;
}

/* ---- UPGMA, hw4.cpp:162-226.  Doubles, same operation order; std::to_string(double) == "%f". */
typedef struct { int size; double height; char *newick; } cluster4;

char *orc4_upgma(const double *dist_in, const char *const *names, size_t n) {
    size_t k = n;
    double *d = (double *)malloc((k > 0 ? k * k : 1) * sizeof(double));
    memcpy(d, dist_in, k * k * sizeof(double));
    cluster4 *cl = (cluster4 *)malloc((k ? k : 1) * sizeof(cluster4));
    for (size_t i = 0; i < k; ++i) {
        cl[i].size = 1;
        cl[i].height = 0.0;
        cl[i].newick = dup4(names[i]);
    }
    while (k > 1) {                                                  /* 170: while (clusters.size() - 1) */
        double best = INFINITY;
        size_t im = 0, jm = 0;
        for (size_t i = 0; i < k; ++i)
            for (size_t j = i + 1; j < k; ++j)
                if (d[i * k + j] < best) { best = d[i * k + j]; im = i; jm = j; }   /* 175-183: first strict minimum */
        cluster4 mg;
        mg.size = cl[im].size + cl[jm].size;
        mg.height = best / 2.0;
        char ha[64], hb[64];
        snprintf(ha, sizeof ha, "%f", fabs(mg.height - cl[im].height));
        snprintf(hb, sizeof hb, "%f", fabs(mg.height - cl[jm].height));
        size_t len = strlen(cl[im].newick) + strlen(cl[jm].newick) + strlen(ha) + strlen(hb) + 8;
        mg.newick = (char *)malloc(len);
        snprintf(mg.newick, len, "(%s:%s,%s:%s)", cl[im].newick, ha, cl[jm].newick, hb);   /* 189-190 */
        size_t nk = k - 1;
        double *nd = (double *)calloc(nk > 0 ? nk * nk : 1, sizeof(double));
        cluster4 *ncl = (cluster4 *)malloc(nk * sizeof(cluster4));
        size_t idx = 0;
        for (size_t i = 0; i < k; ++i) {
            if (i == im || i == jm) continue;
            ncl[idx] = cl[i];
            size_t idx2 = 0;
            for (size_t j = 0; j < k; ++j) {
                if (j == im || j == jm) continue;
                nd[idx * nk + idx2] = d[i * k + j];
                ++idx2;
            }
            const double v = (d[im * k + i] * cl[im].size + d[jm * k + i] * cl[jm].size) / mg.size;   /* 219 */
            nd[idx * nk + (nk - 1)] = v;
            nd[(nk - 1) * nk + idx] = v;
            ++idx;
        }
        ncl[nk - 1] = mg;
        free(cl[im].newick);
        free(cl[jm].newick);
        free(cl);
        free(d);
        cl = ncl;
        d = nd;
        k = nk;
    }
    size_t len = (n ? strlen(cl[0].newick) : 0) + 8;
    char *out = (char *)malloc(len);
    snprintf(out, len, "%s:0.0;", n ? cl[0].newick : "");            /* 228 */
    if (n) free(cl[0].newick);
    free(cl);
    free(d);
    return out;
}

/* ---- main, hw4.cpp:74-240 */
typedef struct { char *id; char *seq; size_t len; } rec4;

int orc4_main(int argc, char **argv) {
    if (argc < 7) {
        fprintf(stderr, "Usage: %s -i <input.fasta> -t <tree.txt> -s <match> <mismatch> <gap>\n", argv[0]);
        return 1;
    }
    const char *in = "input.fasta", *outf = "tree.txt";
    int match = 1, mismatch = -1, gap = -1;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-i") && i + 1 < argc) in = argv[++i];
        else if (!strcmp(argv[i], "-t") && i + 1 < argc) outf = argv[++i];
        else if (!strcmp(argv[i], "-s") && i + 3 < argc) {
            match = atoi(argv[++i]);   /* stoi throws on junk in the reference; fixtures use numbers only */
            mismatch = atoi(argv[++i]);
            gap = atoi(argv[++i]);
        } else {
            fprintf(stderr, "Unknown option: %s\n", argv[i]);
            return 1;
        }
    }
    FILE *f = fopen(in, "rb");
    if (!f) {
        fprintf(stderr, "Error opening input file: %s\n", in);
        return 1;
    }
    rec4 *recs = NULL;
    size_t nrec = 0, cap = 0;
    char *cur_id = NULL, *cur = NULL;
    size_t cur_len = 0, cur_cap = 0;
    char *line = NULL;
    size_t lcap = 0;
    for (;;) {
        size_t ll = 0;
        int c, got = 0;
        while ((c = fgetc(f)) != EOF) {
            got = 1;
            if (c == '\n') break;
            if (ll + 2 > lcap) { lcap = lcap ? lcap * 2 : 256; line = (char *)realloc(line, lcap); }
            line[ll++] = (char)c;
        }
        if (c == EOF && !got) break;
        if (ll == 0) { if (c == EOF) break; continue; }             /* 110-112 */
        if (line[ll - 1] == '\r') --ll;                             /* 113-115 */
        const char first = ll ? line[0] : 0;                        /* line[0] of a now-empty std::string is '\0' */
        if (first == '>') {                                         /* 116 */
            if (cur_id && cur_id[0]) {                              /* 117: !currentId.empty() */
                if (nrec == cap) { cap = cap ? cap * 2 : 16; recs = (rec4 *)realloc(recs, cap * sizeof(rec4)); }
                recs[nrec].id = cur_id;
                recs[nrec].seq = (char *)malloc(cur_len + 1);
                memcpy(recs[nrec].seq, cur, cur_len);
                recs[nrec].seq[cur_len] = 0;
                recs[nrec].len = cur_len;
                ++nrec;
            } else free(cur_id);
            cur_id = (char *)malloc(ll);
            memcpy(cur_id, line + 1, ll - 1);
            cur_id[ll - 1] = 0;
            cur_len = 0;
        } else {
            if (cur_len + ll + 1 > cur_cap) { cur_cap = (cur_len + ll + 1) * 2; cur = (char *)realloc(cur, cur_cap); }
            memcpy(cur + cur_len, line, ll);
            cur_len += ll;
        }
        if (c == EOF) break;
    }
    if (cur_id && cur_id[0]) {                                      /* 133-135 */
        if (nrec == cap) { cap = cap ? cap * 2 : 16; recs = (rec4 *)realloc(recs, cap * sizeof(rec4)); }
        recs[nrec].id = cur_id;
        recs[nrec].seq = (char *)malloc(cur_len + 1);
        memcpy(recs[nrec].seq, cur, cur_len);
        recs[nrec].seq[cur_len] = 0;
        recs[nrec].len = cur_len;
        ++nrec;
    }
    fclose(f);
    double *dist = (double *)calloc(nrec > 0 ? nrec * nrec : 1, sizeof(double));
    for (size_t i = 0; i < nrec; ++i)
        for (size_t j = i + 1; j < nrec; ++j) {                     /* 138-159 */
            const double v = (double)orc4_nw_distance(recs[i].seq, recs[i].len, recs[j].seq, recs[j].len, match, mismatch, gap, NULL);
            dist[j * nrec + i] = v;
            dist[i * nrec + j] = v;
        }
    const char **names = (const char **)malloc((nrec ? nrec : 1) * sizeof(char *));
    for (size_t i = 0; i < nrec; ++i) names[i] = recs[i].id;
    /* with zero sequences the reference indexes clusters[0] of an empty vector (UB); fixtures avoid it */
    char *tree = orc4_upgma(dist, names, nrec);
    FILE *o = fopen(outf, "wb");
    if (!o) {
        fprintf(stderr, "Error opening output file: %s\n", outf);
        return 1;
    }
    fprintf(o, "%s\n", tree);
    fclose(o);
    return 0;
}

#ifdef ORC4_MAIN
int main(int argc, char **argv) { return orc4_main(argc, argv); }
#endif
