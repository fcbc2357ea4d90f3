/* hw3_oracle.c -- TEST INFRASTRUCTURE ONLY (see hw3_oracle.h). */
#include "hw3_oracle.h"

#include <limits.h>
#include <stdlib.h>

#define NEG3 (INT_MIN / 2) /* hw3.cpp:16 INT_MIN_Redefined */

static inline int32_t add3(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }

int32_t orc3_affine_score(const char *s1, size_t n, const char *s2, size_t m, int match, int mismatch, int go, int ge) {
    /* rows of the three matrices for i-1 (prev) and i (cur) */
    int32_t *V0 = (int32_t *)malloc((m + 1) * sizeof(int32_t)), *V1 = (int32_t *)malloc((m + 1) * sizeof(int32_t));
    int32_t *F0 = (int32_t *)malloc((m + 1) * sizeof(int32_t)), *F1 = (int32_t *)malloc((m + 1) * sizeof(int32_t));
    int32_t *E0 = (int32_t *)malloc((m + 1) * sizeof(int32_t)), *E1 = (int32_t *)malloc((m + 1) * sizeof(int32_t));
    V0[0] = 0;                                                       /* 39 */
    F0[0] = E0[0] = NEG3;                                            /* 40 */
    for (size_t j = 1; j <= m; ++j) {                                /* 47-52 */
        V0[j] = NEG3;
        E0[j] = add3(go, (int32_t)((uint32_t)ge * (uint32_t)(j - 1)));
        F0[j] = NEG3;
    }
    for (size_t i = 1; i <= n; ++i) {
        V1[0] = NEG3;                                                /* 42 */
        F1[0] = add3(go, (int32_t)((uint32_t)ge * (uint32_t)(i - 1))); /* 43 */
        E1[0] = NEG3;                                                /* 45 */
        for (size_t j = 1; j <= m; ++j) {                            /* 55-84 */
            const int32_t sub = (s1[i - 1] == s2[j - 1]) ? match : mismatch;
            int32_t v = add3(V0[j - 1], sub);                        /* 59 */
            if (add3(F0[j - 1], sub) > v) v = add3(F0[j - 1], sub);  /* 61-64 */
            if (add3(E0[j - 1], sub) > v) v = add3(E0[j - 1], sub);  /* 65-68 */
            int32_t f = add3(add3(V0[j], go), ge);                   /* 70 */
            if (add3(F0[j], ge) > f) f = add3(F0[j], ge);            /* 72-75 */
            int32_t e = add3(add3(V1[j - 1], go), ge);               /* 77 */
            if (add3(E1[j - 1], ge) > e) e = add3(E1[j - 1], ge);    /* 79-82 */
            V1[j] = v;
            F1[j] = f;
            E1[j] = e;
        }
        int32_t *t;
        t = V0; V0 = V1; V1 = t;
        t = F0; F0 = F1; F1 = t;
        t = E0; E0 = E1; E1 = t;
    }
    int32_t best = V0[m];                                            /* 88-97 */
    if (F0[m] > best) best = F0[m];
    if (E0[m] > best) best = E0[m];
    free(V0); free(V1); free(F0); free(F1); free(E0); free(E1);
    return best;
}

size_t orc3_center(const int32_t *ps, size_t n_seq, int64_t *sum) {
    for (size_t i = 0; i < n_seq; ++i) sum[i] = 0;
    size_t k = 0;
    for (size_t i = 0; i < n_seq; ++i)                               /* 232-241 */
        for (size_t j = i + 1; j < n_seq; ++j, ++k) {
            sum[i] += ps[k];
            sum[j] += ps[k];
        }
    size_t c = 0;                                                    /* 244-251: first strict maximum */
    for (size_t i = 1; i < n_seq; ++i)
        if (sum[i] > sum[c]) c = i;
    return c;
}

/* ------------------------------------------------------------------ alignment with traceback (hw3.cpp:23-135) */
#include <stdio.h>
#include <string.h>

orc3_alignment *orc3_affine_align(const char *s1, size_t n, const char *s2, size_t m, int match, int mismatch, int go, int ge) {
    const size_t W = m + 1, cells = (n + 1) * W;
    int32_t *V = (int32_t *)malloc(cells * sizeof(int32_t)), *F = (int32_t *)malloc(cells * sizeof(int32_t)),
            *E = (int32_t *)malloc(cells * sizeof(int32_t));
    signed char *tV = (signed char *)malloc(cells), *tF = (signed char *)malloc(cells), *tE = (signed char *)malloc(cells);
    orc3_alignment *out = (orc3_alignment *)calloc(1, sizeof *out);
    if (!V || !F || !E || !tV || !tF || !tE || !out) {
        free(V); free(F); free(E); free(tV); free(tF); free(tE); free(out);
        return NULL;
    }
    for (size_t c = 0; c < cells; ++c) {                              /* 28-37 */
        V[c] = F[c] = E[c] = NEG3;
        tV[c] = tF[c] = tE[c] = -1;
    }
    V[0] = 0;                                                        /* 40-41 */
    for (size_t i = 1; i <= n; ++i) {                                /* 42-47 */
        F[i * W] = add3(go, (int32_t)((uint32_t)ge * (uint32_t)(i - 1)));
        tF[i * W] = (i == 1) ? 0 : 1;
    }
    for (size_t j = 1; j <= m; ++j) {                                /* 48-53 */
        E[j] = add3(go, (int32_t)((uint32_t)ge * (uint32_t)(j - 1)));
        tE[j] = (j == 1) ? 0 : 1;
    }
    for (size_t i = 1; i <= n; ++i)
        for (size_t j = 1; j <= m; ++j) {                            /* 55-84 */
            const int32_t sub = (s1[i - 1] == s2[j - 1]) ? match : mismatch;
            const size_t c = i * W + j, d = (i - 1) * W + (j - 1), u = (i - 1) * W + j, l = i * W + (j - 1);
            V[c] = add3(V[d], sub);
            tV[c] = 0;
            if (add3(F[d], sub) > V[c]) { V[c] = add3(F[d], sub); tV[c] = 1; }
            if (add3(E[d], sub) > V[c]) { V[c] = add3(E[d], sub); tV[c] = 2; }
            F[c] = add3(add3(V[u], go), ge);
            tF[c] = 0;
            if (add3(F[u], ge) > F[c]) { F[c] = add3(F[u], ge); tF[c] = 1; }
            E[c] = add3(add3(V[l], go), ge);
            tE[c] = 0;
            if (add3(E[l], ge) > E[c]) { E[c] = add3(E[l], ge); tE[c] = 1; }
        }
    int state = 0;                                                   /* 86-97 */
    int32_t best = V[n * W + m];
    if (F[n * W + m] > best) { best = F[n * W + m]; state = 1; }
    if (E[n * W + m] > best) { best = E[n * W + m]; state = 2; }
    out->score = best;
    out->ops = (char *)malloc(n + m + 1);
    size_t i = n, j = m, k = 0;
    while (i > 0 || j > 0) {                                         /* 105-131 */
        if (state == 0) {
            const int prev = tV[i * W + j];
            out->ops[k++] = 'M';
            --i; --j;
            state = prev;
        } else if (state == 1) {
            state = (tF[i * W + j] == 0) ? 0 : 1;
            out->ops[k++] = 'D';
            --i;
        } else {
            state = (tE[i * W + j] == 0) ? 0 : 2;
            out->ops[k++] = 'I';
            --j;
        }
    }
    out->ops[k] = 0;
    out->len = k;
    out->a1 = (char *)malloc(k + 1);
    out->a2 = (char *)malloc(k + 1);
    size_t p1 = 0, p2 = 0;                                           /* forward strings (133-134) */
    for (size_t c = 0; c < k; ++c) {
        const char op = out->ops[k - 1 - c];
        out->a1[c] = (op == 'I') ? '-' : s1[p1++];
        out->a2[c] = (op == 'D') ? '-' : s2[p2++];
    }
    out->a1[k] = out->a2[k] = 0;
    free(V); free(F); free(E); free(tV); free(tF); free(tE);
    return out;
}

void orc3_free_alignment(orc3_alignment *a) {
    if (!a) return;
    free(a->a1); free(a->a2); free(a->ops); free(a);
}

/* ------------------------------------------------------------------ readFASTA (hw3.cpp:137-167) */
static int isspace3(int c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

typedef struct { char *p; size_t len, cap; } buf3;
static void b3_put(buf3 *b, const char *s, size_t n) {
    if (b->len + n + 1 > b->cap) {
        b->cap = (b->len + n + 1) * 2;
        b->p = (char *)realloc(b->p, b->cap);
    }
    memcpy(b->p + b->len, s, n);
    b->len += n;
    b->p[b->len] = 0;
}

static void f3_push(orc3_fasta *f, size_t *cap, const buf3 *header, const buf3 *seq) {
    if (f->count == *cap) {
        *cap = *cap ? *cap * 2 : 16;
        f->header = (char **)realloc(f->header, *cap * sizeof(char *));
        f->seq = (char **)realloc(f->seq, *cap * sizeof(char *));
        f->len = (size_t *)realloc(f->len, *cap * sizeof(size_t));
    }
    f->header[f->count] = (char *)malloc(header->len + 1);
    memcpy(f->header[f->count], header->len ? header->p : "", header->len);
    f->header[f->count][header->len] = 0;
    f->seq[f->count] = (char *)malloc(seq->len + 1);
    memcpy(f->seq[f->count], seq->len ? seq->p : "", seq->len);
    f->seq[f->count][seq->len] = 0;
    f->len[f->count] = seq->len;
    ++f->count;
}

int orc3_read_fasta(const char *path, orc3_fasta *out) {
    memset(out, 0, sizeof *out);
    FILE *fp = fopen(path, "rb");
    if (!fp) return -1;                                              /* 139-142 */
    size_t cap = 0;
    buf3 line = {0, 0, 0}, header = {0, 0, 0}, seq = {0, 0, 0};
    int eof = 0;
    while (!eof) {                                                   /* getline loop, 146 */
        line.len = 0;
        int got = 0, c;
        while ((c = fgetc(fp)) != EOF) {
            got = 1;
            if (c == '\n') break;
            const char ch = (char)c;
            b3_put(&line, &ch, 1);
        }
        if (c == EOF) {
            eof = 1;
            if (!got) break;
        }
        if (line.len == 0) continue;                                 /* 147-149 */
        if (line.p[0] == '>') {                                      /* 150-155 */
            if (header.len != 0) {
                f3_push(out, &cap, &header, &seq);
                seq.len = 0;
            }
            header.len = 0;
            b3_put(&header, line.p + 1, line.len - 1);
        } else {
            for (size_t k = 0; k < line.len; ++k)                    /* 157-160 */
                if (!isspace3((unsigned char)line.p[k])) b3_put(&seq, line.p + k, 1);
        }
    }
    if (header.len != 0) f3_push(out, &cap, &header, &seq);          /* 163-165 */
    fclose(fp);
    free(line.p); free(header.p); free(seq.p);
    return 0;
}

void orc3_free_fasta(orc3_fasta *f) {
    for (size_t i = 0; i < f->count; ++i) {
        free(f->header[i]);
        free(f->seq[i]);
    }
    free(f->header); free(f->seq); free(f->len);
    memset(f, 0, sizeof *f);
}

/* ------------------------------------------------------------------ main (hw3.cpp:169-368) */
int orc3_hw3_main(int argc, char **argv) {
    if (argc < 7) {                                                  /* 170-173 */
        printf("Usage: %s -i input.fasta -o output.phy -s matchScore:mismatchScore:gapOpeningScore:gapExtensionScore\n", argv[0]);
        return 0;
    }
    const char *in_file = "", *out_file = "", *scores = "";
    for (int i = 1; i < argc; ++i) {                                 /* 176-188 */
        if (!strcmp(argv[i], "-i") && i + 1 < argc) in_file = argv[++i];
        else if (!strcmp(argv[i], "-o") && i + 1 < argc) out_file = argv[++i];
        else if (!strcmp(argv[i], "-s") && i + 1 < argc) scores = argv[++i];
        else {
            printf("Unknown argument: %s\n", argv[i]);
            return 0;
        }
    }
    int val[4], nv = 0;                                              /* 191-207: tokens split at ':' */
    {
        const char *p = scores;
        while (*p) {
            const char *q = strchr(p, ':');
            const size_t tl = q ? (size_t)(q - p) : strlen(p);
            char tok[64];
            if (tl >= sizeof tok) return 134;
            memcpy(tok, p, tl);
            tok[tl] = 0;
            char *end = NULL;
            const long v = strtol(tok, &end, 10);                    /* std::stoi: leading blanks, sign, digits */
            if (end == tok) return 134;                              /* stoi throws -> terminate */
            if (nv < 4) val[nv] = (int)v;
            ++nv;
            if (!q) break;
            p = q + 1;
        }
        if (nv != 4) {
            printf("Error: Score must have four values separated by ':'\n");
            return 0;
        }
    }
    const int match = val[0], mismatch = val[1], go = val[2], ge = val[3];
    orc3_fasta fa;
    if (orc3_read_fasta(in_file, &fa) != 0) {                        /* 139-142 */
        printf("Error: Could not open file %s\n", in_file);
        return 1;
    }
    const size_t N = fa.count;
    if (N == 0) {                                                    /* 217-220 */
        printf("No sequences found in %s\n", in_file);
        orc3_free_fasta(&fa);
        return 0;
    }
    if (N == 1) {                                                    /* 223-230 */
        FILE *o = fopen(out_file, "wb");
        if (o) {
            fprintf(o, "1 %zu\n%s %s\n", fa.len[0], fa.header[0], fa.seq[0]);
            fclose(o);
        }
        orc3_free_fasta(&fa);
        return 0;
    }
    int *sum = (int *)calloc(N, sizeof(int));                        /* 233-243 (int sums, as the reference) */
    for (size_t i = 0; i < N; ++i)
        for (size_t j = i + 1; j < N; ++j) {
            const int32_t s = orc3_affine_score(fa.seq[i], fa.len[i], fa.seq[j], fa.len[j], match, mismatch, go, ge);
            sum[i] = add3(sum[i], s);
            sum[j] = add3(sum[j], s);
        }
    size_t c = 0;                                                    /* 246-253 */
    for (size_t i = 1; i < N; ++i)
        if (sum[i] > sum[c]) c = i;
    free(sum);
    const size_t L = fa.len[c];
    int **gap = (int **)calloc(N, sizeof(int *));                    /* 256-283 */
    orc3_alignment **al = (orc3_alignment **)calloc(N, sizeof *al);
    for (size_t i = 0; i < N; ++i) gap[i] = (int *)calloc(L + 1, sizeof(int));
    for (size_t i = 0; i < N; ++i) {
        if (i == c) continue;
        al[i] = orc3_affine_align(fa.seq[c], L, fa.seq[i], fa.len[i], match, mismatch, go, ge);
        size_t pos = 0;
        for (size_t k = 0; k < al[i]->len; ++k) {
            if (al[i]->a1[k] == '-') ++gap[i][pos];
            else ++pos;
        }
    }
    int *merged = (int *)calloc(L + 1, sizeof(int));                 /* 286-291 */
    for (size_t i = 0; i < N; ++i)
        for (size_t k = 0; k <= L; ++k)
            if (gap[i][k] > merged[k]) merged[k] = gap[i][k];
    buf3 *fin = (buf3 *)calloc(N, sizeof(buf3));
    for (size_t k = 0; k <= L; ++k) {                                /* 296-301 */
        for (int g = 0; g < merged[k] - gap[c][k]; ++g) b3_put(&fin[c], "-", 1);
        if (k < L) b3_put(&fin[c], fa.seq[c] + k, 1);
    }
    if (!fin[c].p) b3_put(&fin[c], "", 0);
    for (size_t i = 0; i < N; ++i) {                                 /* 303-328 */
        if (i == c) continue;
        b3_put(&fin[i], "", 0);
        size_t fi = 0, oi = 0;
        while (fi < fin[c].len && oi < al[i]->len) {
            if (fin[c].p[fi] == al[i]->a1[oi]) {
                b3_put(&fin[i], al[i]->a2 + oi, 1);
                ++fi; ++oi;
            } else if (fin[c].p[fi] == '-') {
                b3_put(&fin[i], "-", 1);
                ++fi;
            } else {
                b3_put(&fin[i], al[i]->a2 + oi, 1);
                ++oi;
            }
        }
        while (fi < fin[c].len) {
            b3_put(&fin[i], "-", 1);
            ++fi;
        }
    }
    {                                                                /* 331-332: center first */
        buf3 t = fin[0]; fin[0] = fin[c]; fin[c] = t;
        char *h = fa.header[0]; fa.header[0] = fa.header[c]; fa.header[c] = h;
    }
    FILE *o = fopen(out_file, "wb");                                 /* 334-338 */
    if (!o) {
        printf("Error: Could not open output file %s\n", out_file);
    } else {
        fprintf(o, "%zu %zu\n", N, fin[0].len);                      /* 340 */
        for (size_t i = 0; i < N; ++i) {                             /* 341-356 */
            const size_t hl = strlen(fa.header[i]);
            for (size_t k = 0; k < 10; ++k) fputc(k < hl ? fa.header[i][k] : ' ', o);
            fputc(fin[i].len ? fin[i].p[0] : 0, o);                  /* 349: operator[](0) of an empty string is its terminator */
            for (size_t j = 1; j < fin[i].len; ++j) {
                if (j % 10 == 0) fputc(' ', o);
                fputc(fin[i].p[j], o);
            }
            fputc('\n', o);
        }
        fclose(o);
    }
    for (size_t i = 0; i < N; ++i) {
        free(gap[i]);
        free(fin[i].p);
        orc3_free_alignment(al[i]);
    }
    free(gap); free(al); free(merged); free(fin);
    orc3_free_fasta(&fa);
    return 0;
}

#ifdef ORC3_MAIN
int main(int argc, char **argv) { return orc3_hw3_main(argc, argv); }
#endif
