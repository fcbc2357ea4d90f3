// batch_affine.hip.h -- scores-only AFFINE-gap global alignment for many pairs (gfx950 / MI355X).
//
// Replaces the score pass of hw3.cpp's all-pairs loop (Multiple_Sequence_Alignment/hw3.cpp:232-241):
//   affine_alignment(Si, Sj, M, Mm, Go, Ge, &score)  with only the score requested (hw3.cpp:23-102)
//   V = max(V,F,E)[i-1][j-1] + s          (59-68)      boundary V[0][0] = 0, else INT_MIN/2   (39-52)
//   F = max(V[i-1][j] + Go + Ge, F[i-1][j] + Ge)  (70-75)   F[i][0] = Go + Ge(i-1)
//   E = max(V[i][j-1] + Go + Ge, E[i][j-1] + Ge)  (77-82)   E[0][j] = Go + Ge(j-1)
//   score = max(V, F, E)[n][m]            (88-97)
// Note what the recurrence does NOT allow (and a textbook Gotoh kernel would get wrong): F never
// reads E and E never reads F, a boundary gap of length L costs Go + Ge(L-1) but an interior one
// Go + Ge*L.
//
// Same mapping as batch_scores.hip.h (lane = pair, text symbol in an SGPR, register strips of R rows,
// 4 skewed columns per block, branch-free strip hand-off through HBM, atomic task queue), with three
// values per row instead of one: D = max(V,F,E), E, and Vg = V + Go(+Ge).  The strip hand-off carries
// two values per column: D of the bottom row and the F the next row will see.
//
// SHIFT = true works in coordinates shifted by Ge(i+j):  V~ = D~diag + (s - 2Ge),
// F~ = max(V~up + Go, F~up), E~ = max(V~left + Go, E~left): 5 VALU per cell + 0.5 for the table
// (add_sdwa, add, max, max, max3).  It is used when every value stays far inside int32 (host check);
// otherwise the plain form (7 + 0.5).  The INT_MIN/2 sentinels only ever lose a max against finite
// values, so their exact value is immaterial for n, m >= 1 (empty sides are resolved on the host).
#pragma once
#include "batch_scores.hip.h"

namespace pwa {

struct AffineParams {
    BatchParams b;          // arena, tasks, slots, scores, hand, queue, tables (match/mismatch/pad as there)
    int32_t go, ge;         // gap open / gap extension (hw3.cpp:25)
    int32_t neg;            // "minus infinity"
};

template <int R, int C, int SCORE, bool SHIFT>
__device__ __forceinline__ void affine_block(int (&Vg)[R], int (&E)[R], int (&D)[R], const uint32_t (&pk)[R / 4],
                                             const uint32_t (&cs)[C], const int (&dtop)[C], const int (&fin)[C], int& topprev,
                                             int (&dbot)[C], int (&fbot)[C], const AffineParams& P) {
    constexpr int Q = R / 4;
    int d[C], vu[C], fu[C], dl[C];
#pragma unroll
    for (int k = 0; k < C; ++k) {
        d[k] = (k == 0) ? topprev : dtop[k - 1];   // D of the row above, previous column
        vu[k] = fin[k];                            // makes the first row's F equal to the handed-in F
        fu[k] = SHIFT ? fin[k] : P.neg;
        dl[k] = 0;
    }
    topprev = dtop[C - 1];
    const int ge = P.ge, vadd = SHIFT ? P.go : addw(P.go, P.ge);
#pragma unroll
    for (int step = 0; step < Q + C - 1; ++step) {
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int q = step - k;
            if (q >= 0 && q < Q) {
                uint32_t s4 = 0;
                if (SCORE == SC_PERM) s4 = __builtin_amdgcn_perm(P.b.tab_hi, P.b.tab_lo, pk[q] ^ cs[k]);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int r = 4 * q + b;
                    int s;
                    if (SCORE == SC_PERM) s = (int)(int8_t)(s4 >> (8 * b));
                    else s = (((pk[q] >> (8 * b)) & 0xffu) == cs[k]) ? P.b.match : P.b.mismatch;
                    const int v = addw(d[k], s);                                             // hw3.cpp:59-68
                    d[k] = D[r];
                    const int f = SHIFT ? max(vu[k], fu[k]) : max(vu[k], addw(fu[k], ge));   // hw3.cpp:70-75
                    const int e = SHIFT ? max(Vg[r], E[r]) : max(Vg[r], addw(E[r], ge));     // hw3.cpp:77-82
                    const int vg = addw(v, vadd);
                    const int dn = max(max(v, f), e);                                        // read by (i+1, j+1)
                    Vg[r] = vg;
                    E[r] = e;
                    D[r] = dn;
                    vu[k] = vg;
                    fu[k] = f;
                    dl[k] = dn;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < C; ++k) {
        dbot[k] = dl[k];                                                      // D of the strip's bottom row
        fbot[k] = SHIFT ? max(vu[k], fu[k]) : max(vu[k], addw(fu[k], ge));    // F of the row below it
    }
}

constexpr int affine_waves_per_simd(int R) { return R > 40 ? 2 : 3; }

template <int R, int SCORE, bool SHIFT>
__global__ __launch_bounds__(64, affine_waves_per_simd(R)) void batch_affine_kernel(const AffineParams P) {
    constexpr int Q = R / 4;
    const BatchParams& B = P.b;
    const int lane = threadIdx.x;
    int32_t* const hand = B.hand + (size_t)blockIdx.x * B.hand_stride;
    const int go = P.go, ge = P.ge, neg = P.neg;

    for (;;) {
        uint32_t tid = 0;
        {
            int elect = lane;   // opaque electing lane: see batch_scores.hip.h
            asm volatile("" : "+v"(elect));
            if (elect == 0) tid = atomicAdd(B.queue, 1u);
        }
        tid = __builtin_amdgcn_readfirstlane(tid);
        if (tid >= B.n_tasks) break;

        const BatchTask task = B.tasks[tid];
        const int m = (int)task.text_len;
        const uint32_t* tx = reinterpret_cast<const uint32_t*>(B.arena + task.text_off);
        const uint32_t slot = task.slot0 + lane;
        const uint32_t poff = B.slot_poff[slot];
        const int n = (int)B.slot_plen[slot];
        const uint32_t outi = B.slot_out[slot];
        const int nblk = m >> 2, rem = m & 3;
        int result = 0;

        for (int s = 0; s < (int)task.n_strips; ++s) {
            const int row0 = s * R;
            uint32_t pk[Q];
            {
                const uint32_t* pp = reinterpret_cast<const uint32_t*>(B.arena + poff + row0);
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const int valid = n - (row0 + 4 * q);
                    const uint32_t w = pp[q];
                    const uint32_t keep = valid >= 4 ? 0xffffffffu : (valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u));
                    pk[q] = (w & keep) | (B.pad_word & ~keep);
                }
            }
            // ---- column 0 (hw3.cpp:41-46): V = E = -inf, D = F = Go + Ge(i-1); shifted by Ge*i: Go - Ge
            int Vg[R], E[R], D[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                Vg[r] = neg;
                E[r] = neg;
                D[r] = SHIFT ? addw(go, -ge) : addw(go, mulw(row0 + r, ge));
            }
            // D of the row above the strip at column 0: V[0][0] = 0 (hw3.cpp:39) or that row's column-0 value
            int topprev = (s == 0) ? 0 : (SHIFT ? addw(go, -ge) : addw(go, mulw(row0 - 1, ge)));

            const bool has_top = s > 0;
            const bool has_bot = s + 1 < (int)task.n_strips;
            const int32_t* hin = hand + (size_t)((s + 1) & 1) * B.hand_half;
            int32_t* hout = hand + (size_t)(s & 1) * B.hand_half;
            const size_t in_stride = has_top ? 128 : 0, out_stride = has_bot ? 128 : 0;   // two int4 per lane per block
            const int4* hin4 = reinterpret_cast<const int4*>(hin) + lane;
            int4* hout4 = reinterpret_cast<int4*>(hout) + lane;
            int4 dnext = hin4[0], fnext = hin4[64];
            uint32_t cwn = tx[0];
            for (int jb = 0; jb < nblk; ++jb) {
                const uint32_t cw = cwn;
                const int4 dcur = dnext, fcur = fnext;
                cwn = tx[jb + 1];
                dnext = hin4[(size_t)(jb + 1) * in_stride];
                fnext = hin4[(size_t)(jb + 1) * in_stride + 64];
                int dtop[4], fin[4], dbot[4], fbot[4];
                uint32_t cs[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t c = (cw >> (8 * k)) & 0xffu;
                    cs[k] = (SCORE == SC_PERM) ? c * 0x01010101u : c;
                }
                {
                    const int dl[4] = {dcur.x, dcur.y, dcur.z, dcur.w}, fl[4] = {fcur.x, fcur.y, fcur.z, fcur.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // row 0 (hw3.cpp:47-52): D[0][j] = E[0][j] = Go + Ge(j-1) (shifted: Go - Ge), F of row 1 = -inf
                        dtop[k] = has_top ? dl[k] : (SHIFT ? addw(go, -ge) : addw(go, mulw(4 * jb + k, ge)));
                        fin[k] = has_top ? fl[k] : neg;
                    }
                }
                affine_block<R, 4, SCORE, SHIFT>(Vg, E, D, pk, cs, dtop, fin, topprev, dbot, fbot, P);
                hout4[(size_t)jb * out_stride] = make_int4(dbot[0], dbot[1], dbot[2], dbot[3]);
                hout4[(size_t)jb * out_stride + 64] = make_int4(fbot[0], fbot[1], fbot[2], fbot[3]);
            }
            if (rem > 0) {
                uint32_t cw = cwn;
                int d0 = dnext.x, d1 = dnext.y, d2 = dnext.z, f0 = fnext.x, f1 = fnext.y, f2 = fnext.z;
#pragma unroll 1
                for (int k = 0; k < rem; ++k) {
                    const uint32_t c = cw & 0xffu;
                    cw >>= 8;
                    const uint32_t cs1[1] = {(SCORE == SC_PERM) ? c * 0x01010101u : c};
                    const int dtop1[1] = {has_top ? d0 : (SHIFT ? addw(go, -ge) : addw(go, mulw(4 * nblk + k, ge)))};
                    const int fin1[1] = {has_top ? f0 : neg};
                    d0 = d1; d1 = d2;
                    f0 = f1; f1 = f2;
                    int dbot1[1], fbot1[1];
                    affine_block<R, 1, SCORE, SHIFT>(Vg, E, D, pk, cs1, dtop1, fin1, topprev, dbot1, fbot1, P);
                    hout[((size_t)nblk * out_stride + lane) * 4 + k] = dbot1[0];
                    hout[((size_t)nblk * out_stride + 64 + lane) * 4 + k] = fbot1[0];
                }
            }
            // ---- max(V, F, E)[n][m] sits in this strip for the lanes whose pattern ends here (hw3.cpp:88-97)
            {
                const int rl = n - 1 - row0;
                if (rl >= 0 && rl < R) {
                    int v = 0;
#pragma unroll
                    for (int r = 0; r < R; ++r) v = (rl == r) ? D[r] : v;
                    result = v;
                }
            }
        }
        if (outi != 0xffffffffu) B.scores[outi] = SHIFT ? addw(result, mulw(n + m, ge)) : result;
    }
}

}  // namespace pwa
