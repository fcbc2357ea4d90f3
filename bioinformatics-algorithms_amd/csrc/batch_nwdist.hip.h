// batch_nwdist.hip.h -- hw4's all-pairs step: NW with hw4's tie-break + the traceback-derived distance,
// without ever storing a traceback (gfx950 / MI355X).
//
// Replaces, per pair, hw4/hw4.cpp:16-72 (needleman_wunsch: v = diag + s; if (up > v) up; if (left > v)
// left -- i.e. diag >= up >= left, NOT hw2's diag >= left >= up) followed by the distance rule of
// main (hw4.cpp:146-152): number of alignment columns that hold a gap or a mismatch.
// The walk from (n, m) follows exactly one pointer per cell, so the distance of the path through a cell
// is a second DP value carried along the chosen predecessor:
//     dist[i][j] = dist[pred] + (diag ? (s1[i-1] != s2[j-1]) : 1),   dist[i][0] = i, dist[0][j] = j
// and the answer is dist[n][m].  Same mapping as batch_scores.hip.h (lane = pair, text symbol in an SGPR,
// register strips, 4 skewed columns per block, branch-free two-value strip hand-off, atomic task queue).
// Stored per row: Hg = H + gap and Dp = dist + 1, the forms both consumers (the cell below and the cell
// to the right) need; the diagonal consumer gets the -gap / -1 folded into the two byte tables.
// VALU per cell: 2 sdwa adds, 2 x (cmp, max, cndmask), 2 adds, 3/4 for the tables = 10.75 (SC_PERM).
#pragma once
#include "batch_scores.hip.h"

namespace pwa {

struct NwDistParams {
    BatchParams b;            // match/mismatch hold (s - gap) here; tab_hi/tab_lo the same as bytes
    uint32_t tab2_hi, tab2_lo;// selector 0 -> 0xFF (-1: match costs nothing), 1..7 -> 0 (mismatch: +1 stays)
    int32_t* scores2;         // optional NW score per pair (same order), may be nullptr
};

template <int R, int C, int SCORE>
__device__ __forceinline__ void nwdist_block(int (&Hg)[R], int (&Dp)[R], const uint32_t (&pk)[R / 4], const uint32_t (&cs)[C],
                                             const int (&htop)[C], const int (&dtop)[C], int& hprev, int& dprev,
                                             int (&hbot)[C], int (&dbot)[C], const NwDistParams& P) {
    constexpr int Q = R / 4;
    int dh[C], dd[C], uh[C], ud[C];
#pragma unroll
    for (int k = 0; k < C; ++k) {
        dh[k] = (k == 0) ? hprev : htop[k - 1];   // (H + gap, dist + 1) of the row above, previous column
        dd[k] = (k == 0) ? dprev : dtop[k - 1];
        uh[k] = htop[k];
        ud[k] = dtop[k];
    }
    hprev = htop[C - 1];
    dprev = dtop[C - 1];
    const int gap = P.b.gap;
#pragma unroll
    for (int step = 0; step < Q + C - 1; ++step) {
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int q = step - k;
            if (q >= 0 && q < Q) {
                uint32_t s4 = 0, e4 = 0;
                if (SCORE == SC_PERM) {
                    const uint32_t x = pk[q] ^ cs[k];
                    s4 = __builtin_amdgcn_perm(P.b.tab_hi, P.b.tab_lo, x);
                    e4 = __builtin_amdgcn_perm(P.tab2_hi, P.tab2_lo, x);
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int r = 4 * q + b;
                    int s, e;
                    if (SCORE == SC_PERM) {
                        s = (int)(int8_t)(s4 >> (8 * b));
                        e = (int)(int8_t)(e4 >> (8 * b));
                    } else {
                        const bool eq = ((pk[q] >> (8 * b)) & 0xffu) == cs[k];
                        s = eq ? P.b.match : P.b.mismatch;
                        e = eq ? -1 : 0;
                    }
                    int h = addw(dh[k], s);       // diag: H[i-1][j-1] + s          (hw4.cpp:36)
                    int d = addw(dd[k], e);       //       dist + (mismatch ? 1 : 0)
                    const int lh = Hg[r], ld = Dp[r];
                    dh[k] = lh;
                    dd[k] = ld;
                    if (uh[k] > h) { h = uh[k]; d = ud[k]; }   // up first   (hw4.cpp:40-43)
                    if (lh > h) { h = lh; d = ld; }           // then left  (hw4.cpp:44-47)
                    const int hg = addw(h, gap), dp = addw(d, 1);
                    Hg[r] = hg;
                    Dp[r] = dp;
                    uh[k] = hg;
                    ud[k] = dp;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < C; ++k) {
        hbot[k] = uh[k];
        dbot[k] = ud[k];
    }
}

constexpr int nwdist_waves_per_simd(int R) { return R > 60 ? 2 : 3; }

template <int R, int SCORE>
__global__ __launch_bounds__(64, nwdist_waves_per_simd(R)) void batch_nwdist_kernel(const NwDistParams P) {
    constexpr int Q = R / 4;
    const BatchParams& B = P.b;
    const int lane = threadIdx.x;
    int32_t* const hand = B.hand + (size_t)blockIdx.x * B.hand_stride;
    const int gap = B.gap;

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
        int res_h = 0, res_d = 0;

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
            // ---- column 0 (hw4.cpp:21-24): H[i][0] = i*gap, dist = i  ->  Hg = (i+1)*gap, Dp = i+1
            int Hg[R], Dp[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                Hg[r] = mulw(row0 + r + 2, gap);
                Dp[r] = row0 + r + 2;
            }
            int hprev = mulw(row0 + 1, gap), dprev = row0 + 1;   // row above the strip, column 0

            const bool has_top = s > 0;
            const bool has_bot = s + 1 < (int)task.n_strips;
            const int32_t* hin = hand + (size_t)((s + 1) & 1) * B.hand_half;
            int32_t* hout = hand + (size_t)(s & 1) * B.hand_half;
            const size_t in_stride = has_top ? 128 : 0, out_stride = has_bot ? 128 : 0;   // two int4 per lane per block
            const int4* hin4 = reinterpret_cast<const int4*>(hin) + lane;
            int4* hout4 = reinterpret_cast<int4*>(hout) + lane;
            int4 hnext = hin4[0], dnext = hin4[64];
            uint32_t cwn = tx[0];
            for (int jb = 0; jb < nblk; ++jb) {
                const uint32_t cw = cwn;
                const int4 hcur = hnext, dcur = dnext;
                cwn = tx[jb + 1];
                hnext = hin4[(size_t)(jb + 1) * in_stride];
                dnext = hin4[(size_t)(jb + 1) * in_stride + 64];
                int htop[4], dtop[4], hbot[4], dbot[4];
                uint32_t cs[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t c = (cw >> (8 * k)) & 0xffu;
                    cs[k] = (SCORE == SC_PERM) ? c * 0x01010101u : c;
                }
                {
                    const int hl[4] = {hcur.x, hcur.y, hcur.z, hcur.w}, dl[4] = {dcur.x, dcur.y, dcur.z, dcur.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        // row 0 (hw4.cpp:25-28): H[0][j] = j*gap, dist = j  ->  (j+1)*gap, j+1
                        htop[k] = has_top ? hl[k] : mulw(4 * jb + k + 2, gap);
                        dtop[k] = has_top ? dl[k] : 4 * jb + k + 2;
                    }
                }
                nwdist_block<R, 4, SCORE>(Hg, Dp, pk, cs, htop, dtop, hprev, dprev, hbot, dbot, P);
                hout4[(size_t)jb * out_stride] = make_int4(hbot[0], hbot[1], hbot[2], hbot[3]);
                hout4[(size_t)jb * out_stride + 64] = make_int4(dbot[0], dbot[1], dbot[2], dbot[3]);
            }
            if (rem > 0) {
                uint32_t cw = cwn;
                int h0 = hnext.x, h1 = hnext.y, h2 = hnext.z, d0 = dnext.x, d1 = dnext.y, d2 = dnext.z;
#pragma unroll 1
                for (int k = 0; k < rem; ++k) {
                    const uint32_t c = cw & 0xffu;
                    cw >>= 8;
                    const uint32_t cs1[1] = {(SCORE == SC_PERM) ? c * 0x01010101u : c};
                    const int htop1[1] = {has_top ? h0 : mulw(4 * nblk + k + 2, gap)};
                    const int dtop1[1] = {has_top ? d0 : 4 * nblk + k + 2};
                    h0 = h1; h1 = h2;
                    d0 = d1; d1 = d2;
                    int hbot1[1], dbot1[1];
                    nwdist_block<R, 1, SCORE>(Hg, Dp, pk, cs1, htop1, dtop1, hprev, dprev, hbot1, dbot1, P);
                    hout[((size_t)nblk * out_stride + lane) * 4 + k] = hbot1[0];
                    hout[((size_t)nblk * out_stride + 64 + lane) * 4 + k] = dbot1[0];
                }
            }
            // ---- (H, dist)[n][m] sits in this strip for the lanes whose pattern ends here
            {
                const int rl = n - 1 - row0;
                if (rl >= 0 && rl < R) {
                    int vh = 0, vd = 0;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        vh = (rl == r) ? Hg[r] : vh;
                        vd = (rl == r) ? Dp[r] : vd;
                    }
                    res_h = vh;
                    res_d = vd;
                }
            }
        }
        if (outi != 0xffffffffu) {
            B.scores[outi] = res_d - 1;                               // hw4.cpp:146-152
            if (P.scores2) P.scores2[outi] = addw(res_h, -gap);      // dp[n][m]
        }
    }
}

}  // namespace pwa
