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
