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



// ---------------------------------------------------------------------------------------------------
// Packed form: (H, dist) of a cell as ONE int32 key, so that one v_max3 both picks hw4's predecessor and carries
// its distance along:
//     K = H * 2^14  +  prio * 2^12  +  dist,      prio: diag 2, up 1, left 0  (hw4.cpp:36-47: diag >= up >= left)
// Keys compare by H first, then by the direction's priority; dist sits below and can never decide (two candidates
// of one cell always differ in prio).  Stored per row: L = key without prio + (gap * 2^14 + 1), i.e. the candidate
// the cell to the RIGHT sees; the cell BELOW adds 2^12 to it (prio 1); the DIAGONAL consumer adds
// ((s - gap) * 2^14 + 2 * 2^12 + e - 1), one of two constants picked by the symbol compare.
// Per cell: cmp, cndmask, add (diag), add (up), max3, and (strip prio), add = 7 VALU against 10.75, one hand-off
// value per column instead of two.  Valid while dist < 4094 and |H| stays inside 17 bits (host check); the plain
// form above takes everything else.
constexpr int kDistBits = 12, kPrioShift = 12, kHShift = 14;

template <int R, int C>
__device__ __forceinline__ void nwdist_packed_block(int (&L)[R], const uint32_t (&pk)[R / 4], const uint32_t (&cs)[C],
                                                    const int (&ltop)[C], int& lprev, int (&lbot)[C], int cd_match, int cd_mis,
                                                    int cl) {
    constexpr int Q = R / 4;
    int dk[C], uk[C];
#pragma unroll
    for (int k = 0; k < C; ++k) {
        dk[k] = (k == 0) ? lprev : ltop[k - 1];   // L of the row above, previous column
        uk[k] = ltop[k];                          // L of the row above, this column
    }
    lprev = ltop[C - 1];
#pragma unroll
    for (int step = 0; step < Q + C - 1; ++step) {
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int q = step - k;
            if (q >= 0 && q < Q) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int r = 4 * q + b;
                    const bool eq = ((pk[q] >> (8 * b)) & 0xffu) == cs[k];
                    const int kd = addw(dk[k], eq ? cd_match : cd_mis);   // diag: H + s, dist + (mismatch), prio 2
                    const int ku = addw(uk[k], 1 << kPrioShift);          // up:   H + gap, dist + 1,        prio 1
                    const int kl = L[r];                                  // left: H + gap, dist + 1,        prio 0
                    dk[k] = kl;
                    const int best = max(kd, max(ku, kl));
                    const int l = addw(best & ~(3 << kPrioShift), cl);
                    L[r] = l;
                    uk[k] = l;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < C; ++k) lbot[k] = uk[k];
}

template <int R>
__global__ __launch_bounds__(64, 2) void batch_nwdist_packed_kernel(const NwDistParams P) {
    constexpr int Q = R / 4;
    const BatchParams& B = P.b;
    const int lane = threadIdx.x;
    int32_t* const hand = B.hand + (size_t)blockIdx.x * B.hand_stride;
    // here B.match / B.mismatch / B.gap are hw4's own three scores (not the (s - gap) tables of the plain form)
    const int gap = B.gap;
    const int cl = addw(mulw(gap, 1 << kHShift), 1);
    const int cd_match = addw(mulw(B.match - gap, 1 << kHShift), (2 << kPrioShift) - 1);
    const int cd_mis = addw(mulw(B.mismatch - gap, 1 << kHShift), (2 << kPrioShift));
    // L of a boundary cell with H = x * gap, dist = x (hw4.cpp:21-28)
    auto boundary = [&](int x) -> int { return addw(addw(mulw(mulw(x, gap), 1 << kHShift), x), cl); };

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
        int res = 0;

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
            int L[R];
#pragma unroll
            for (int r = 0; r < R; ++r) L[r] = boundary(row0 + r + 1);   // column 0
            int lprev = boundary(row0);                                  // row above the strip, column 0

            const bool has_top = s > 0;
            const bool has_bot = s + 1 < (int)task.n_strips;
            const int32_t* hin = hand + (size_t)((s + 1) & 1) * B.hand_half;
            int32_t* hout = hand + (size_t)(s & 1) * B.hand_half;
            const size_t in_stride = has_top ? 64 : 0, out_stride = has_bot ? 64 : 0;   // one int4 per lane per block
            const int4* hin4 = reinterpret_cast<const int4*>(hin) + lane;
            int4* hout4 = reinterpret_cast<int4*>(hout) + lane;
            int4 lnext = hin4[0];
            uint32_t cwn = tx[0];
            for (int jb = 0; jb < nblk; ++jb) {
                const uint32_t cw = cwn;
                const int4 lcur = lnext;
                cwn = tx[jb + 1];
                lnext = hin4[(size_t)(jb + 1) * in_stride];
                int ltop[4], lbot[4];
                uint32_t cs[4];
                const int ll[4] = {lcur.x, lcur.y, lcur.z, lcur.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    cs[k] = (cw >> (8 * k)) & 0xffu;
                    ltop[k] = has_top ? ll[k] : boundary(4 * jb + k + 1);   // row 0
                }
                nwdist_packed_block<R, 4>(L, pk, cs, ltop, lprev, lbot, cd_match, cd_mis, cl);
                hout4[(size_t)jb * out_stride] = make_int4(lbot[0], lbot[1], lbot[2], lbot[3]);
            }
            if (rem > 0) {
                uint32_t cw = cwn;
                int l0 = lnext.x, l1 = lnext.y, l2 = lnext.z;
#pragma unroll 1
                for (int k = 0; k < rem; ++k) {
                    const uint32_t cs1[1] = {cw & 0xffu};
                    cw >>= 8;
                    const int ltop1[1] = {has_top ? l0 : boundary(4 * nblk + k + 1)};
                    l0 = l1;
                    l1 = l2;
                    int lbot1[1];
                    nwdist_packed_block<R, 1>(L, pk, cs1, ltop1, lprev, lbot1, cd_match, cd_mis, cl);
                    hout[((size_t)nblk * out_stride + lane) * 4 + k] = lbot1[0];
                }
            }
            {   // the key of (n, m) sits in this strip for the lanes whose pattern ends here
                const int rl = n - 1 - row0;
                if (rl >= 0 && rl < R) {
                    int v = 0;
#pragma unroll
                    for (int r = 0; r < R; ++r) v = (rl == r) ? L[r] : v;
                    res = v;
                }
            }
        }
        if (outi != 0xffffffffu) {
            const int key = addw(res, -cl);                                        // H * 2^14 + dist
            B.scores[outi] = key & ((1 << kDistBits) - 1);                        // hw4.cpp:146-152
            if (P.scores2) P.scores2[outi] = key >> kHShift;                      // dp[n][m]
        }
    }
}

}  // namespace pwa
