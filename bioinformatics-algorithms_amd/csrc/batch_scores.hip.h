// batch_scores.hip.h -- scores-only DP for MANY pairs (gfx950 / MI355X).
//
// Replaces the score part of hw2.cpp's per-pair calls in the loop 328-338:
//   NW  dp[n][m]           recurrence hw2.cpp:138-156
//   SW  max over all cells recurrence hw2.cpp:205-231
//
// Mapping (MI355X-first, nothing like the reference's row-major scalar loop):
//   * one 64-lane wavefront = one "wave task" = up to 64 PATTERNS against ONE shared TEXT;
//     lane l owns the whole DP matrix of pair (pattern_l, text): no cross-lane traffic at all;
//   * the text character of a column is wave-uniform, so it lives in an SGPR (scalar loads);
//   * each lane keeps a register tile of R consecutive DP rows ("strip"): H[R] int32 in VGPRs plus
//     the strip's R pattern symbols packed 4 per VGPR; a column update is R dependent-free
//     register cells; columns are processed 4 at a time in a skewed (anti-diagonal) order so
//     that four independent dependency chains are in flight per lane;
//   * patterns longer than R rows take several strips; the strip's bottom row is handed to the
//     next strip through a per-workgroup HBM row buffer laid out [column/4][lane][4] so that every
//     access is one coalesced 1 KiB dwordx4 wave access (8 B per R cells per lane).  The column
//     loop is kept ONE basic block: the row load/store is unconditional and a stride of 0 parks
//     it on a 1 KiB dummy block when there is no strip above / below (a uniform branch in that
//     loop makes hipcc double the live H registers and spill);
//   * substitution score of 4 rows at once: one v_xor + one v_perm_b32 byte-table lookup
//     (alphabets of <= 7 symbols, scores in int8), then one SDWA add per cell; the generic
//     raw-byte path does compare + select per cell;
//   * wave tasks are pulled from an atomic queue (heterogeneous task sizes).
//
// Integer recurrences only: VALU-issue bound, no MFMA, HBM traffic ~1e-5 B/cell (C3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lds_sync.hip.h"

namespace pwa {

// NWG: NW in gap-shifted space G = H - g(i+j).
// SWS: SW with a second stored form per row, Hs = max(H + g, 0) (one unsigned saturating subtract).  Both gap
//      candidates then arrive already floored at 0, so  max(0, t, u+g, l+g) = max3(t, Hs_up, Hs_left)  with
//      no explicit 0 and no separate "+ g": 3 VALU per cell instead of 4 (+ 1/2 table + 1/2 running best),
//      at the price of 2R instead of R registers per lane (R <= 96).
enum { BM_SW = 0, BM_NW = 1, BM_NWG = 2, BM_SWS = 6 };
enum { SC_PERM = 0, SC_CMP = 1 };

struct BatchTask {
    uint32_t text_off;   // byte offset of the text symbols in the arena (16-byte aligned)
    uint32_t text_len;   // m
    uint32_t slot0;      // first of this task's 64 lane slots
    uint32_t n_strips;   // ceil(longest pattern of the task / R)
};

struct BatchParams {
    const uint8_t* arena;        // symbols of every sequence, each 16-byte aligned, with slack
    const BatchTask* tasks;
    const uint32_t* slot_poff;   // per slot: arena offset of the pattern
    const uint32_t* slot_plen;   // per slot: pattern length, 0 = empty lane
    const uint32_t* slot_out;    // per slot: index into scores
    const uint32_t* slot_toff;   // LANES kernels (every lane its own text): per slot arena offset / length of the text
    const uint32_t* slot_tlen;
    const uint8_t* lane_text;    // LANES + BM_NWG: right-aligned, front-padded text rows (slot_toff points in here)
    int32_t* scores;
    int32_t* hand;               // strip hand-off rows, one region per workgroup
    uint64_t hand_stride;        // int32 elements per workgroup region (2 halves)
    uint32_t hand_half;          // int32 elements per half
    uint32_t* queue;             // atomic task counter (zeroed before every launch)
    uint32_t n_tasks;
    int32_t match, mismatch, gap;
    uint32_t tab_hi, tab_lo;     // SC_PERM: byte table, selector 0 -> match, 1..7 -> mismatch
    uint32_t pad_word;           // symbol that matches nothing in any text, x4
    uint32_t tpad_word;          // LANES: symbol that matches nothing in any pattern and differs from pad_word, x4
};

__device__ __forceinline__ int addw(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int mulw(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

// One block of C columns over the lane's R-row register strip.
//   H[r]     in:  dp[row0+r+1][j0]        (the column left of the block)
//            out: dp[row0+r+1][j0+C]
//   top[k]   dp[row0][j0+1+k]  (row above the strip), topprev = dp[row0][j0]
//   bot[k]   dp[row0+R][j0+1+k]
__device__ __forceinline__ int usub_sat(int a, int b) {   // max(a - b, 0) for a, b >= 0: v_sub_u32 ... clamp
    return (int)__builtin_elementwise_sub_sat((unsigned)a, (unsigned)b);
}

template <int R, int C, int MODE, int SCORE>
__device__ __forceinline__ void dp_block(int (&H)[R], int (&Hs)[MODE == BM_SWS ? R : 1], const uint32_t (&pk)[R / 4],
                                         const uint32_t (&cs)[C], const int (&top)[C], int& topprev, int (&bot)[C], int& best,
                                         const BatchParams& P) {
    constexpr int Q = R / 4;
    int d[C], u[C], hb[C];
    const int gap = P.gap;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        d[k] = (k == 0) ? topprev : top[k - 1];
        u[k] = (MODE == BM_SWS) ? usub_sat(top[k], -gap) : top[k];
        hb[k] = 0;
    }
    topprev = top[C - 1];
    // skewed order: column k runs one quad (4 rows) behind column k-1
#pragma unroll
    for (int step = 0; step < Q + C - 1; ++step) {
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int q = step - k;
            if (q >= 0 && q < Q) {
                uint32_t s4 = 0;
                if (SCORE == SC_PERM) s4 = __builtin_amdgcn_perm(P.tab_hi, P.tab_lo, pk[q] ^ cs[k]);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int r = 4 * q + b;
                    int s;
                    if (SCORE == SC_PERM) s = (int)(int8_t)(s4 >> (8 * b));
                    else s = (((pk[q] >> (8 * b)) & 0xffu) == cs[k]) ? P.match : P.mismatch;
                    const int t = addw(d[k], s);   // diagonal candidate, hw2.cpp:142 / 208
                    const int l = H[r];            // dp[i][j-1]
                    d[k] = l;
                    int h;
                    if (MODE == BM_SWS) {
                        const int ls = Hs[r];                    // max(dp[i][j-1] + g, 0)
                        h = max(max(t, u[k]), ls);               // = max(0, diag, up, left), hw2.cpp:211
                        best = max(best, h);
                        const int hs = usub_sat(h, -gap);
                        Hs[r] = hs;
                        H[r] = h;
                        u[k] = hs;
                        hb[k] = h;
                        continue;
                    }
                    if (MODE == BM_SW) {
                        const int e = addw(max(u[k], l), gap);   // hw2.cpp:209-210
                        h = max(max(t, e), 0);                   // hw2.cpp:211
                        best = max(best, h);                     // hw2.cpp:225 (value only)
                    } else if (MODE == BM_NW) {
                        const int e = addw(max(u[k], l), gap);   // hw2.cpp:140-141
                        h = max(t, e);                           // hw2.cpp:142-153 (value only)
                    } else {
                        h = max(max(t, u[k]), l);                // gap folded into the coordinates
                    }
                    H[r] = h;
                    u[k] = h;
                    hb[k] = h;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < C; ++k) bot[k] = hb[k];
}

// Register budget: H[R] + R/4 packed symbols + ~30 live temporaries.  The second launch-bound is the
// number of waves per SIMD the allocation must leave room for (512 registers per SIMD lane, AGPRs
// included): without it hipcc parks 12 values in AGPRs at R = 152, the wave allocates 268 registers
// and only ONE wave fits per SIMD -- which halves VALU throughput (one wave issues every 4 cycles).
constexpr int strip_waves_per_simd(int R, int MODE) {
    return (MODE == BM_SWS) ? (R > 48 ? 2 : 3) : (R > 104 ? 2 : (R > 80 ? 3 : 4));
}

// MULTI = false: every task of the launch is a single strip -- the hand-off row accesses are compiled out
// (with them, each wave parks an unconditional 1 KiB load + store per 4 columns on an L2-resident dummy
// block: harmless for speed, but it shows up as ~45 MB of HBM traffic per C3 launch).
//
// LANES = true (local alignment only): the 64 pairs of a task do not share a text -- the shape of the reference's
// own loop, pattern i against reference i (hw2.cpp:328-338).  Every lane streams its own text (one dword per
// 4-column block, prefetched a block ahead), the symbol splat moves from the scalar unit to four v_perm_b32, and
// columns past a lane's own text are fed the symbol that matches nothing: local scores can then only decay, so
// the lane's maximum is already final (the same argument that pads short patterns).  ~+3 % VALU per cell
// against the shared-text form, against up to 64x fewer idle lanes on index-paired lists.
// Launch geometry: workgroups of ONE or of FOUR waves (the host's choice); every wave pulls its own tasks and owns its own hand-off region.
// Four-wave workgroups that each ask for a share of the CU's LDS are how the host gets a BALANCED placement -- the same number of
// waves on every SIMD -- whatever kernel ran before ([gpu, r03] tools/probes/simd_place2.hip: single-wave workgroups launched after
// another kernel double up on some SIMDs and leave others empty).
template <int R, int MODE, int SCORE, bool MULTI, bool LANES = false>
__global__ __launch_bounds__(256, strip_waves_per_simd(R, MODE)) void batch_scores_kernel(const BatchParams P) {
    // Global alignment, gap-shifted form, coded alphabets of <= 4 symbols: the texts of a task are RIGHT-aligned.  The
    // host stores every lane's text as a row of M = 4*ceil(max m / 4) codes, front-padded with code 4, which the
    // table scores like a gap (H-space g, here -g): with g <= 0 such a column reproduces column 0 exactly
    // (H[i][j] = i*g), so a lane's real matrix simply starts p = M - m columns late, every lane ends in the last
    // block, and H[n][m] = G'[n][M] + g(n + M).  Only row 0 differs per lane: H[0][j] = g * max(j - p, 0).
    static_assert(!LANES || MODE == BM_SWS || MODE == BM_SW || MODE == BM_NWG, "per-lane texts: SW forms and gap-shifted NW");
    static_assert(R % 4 == 0, "a strip is whole quads of rows (one packed symbol word each)");
    constexpr int Q = R / 4;
    const int lane = threadIdx.x & 63;
    int32_t* const hand = P.hand + ((size_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * P.hand_stride;

    for (;;) {
        uint32_t tid = 0;
        {
            // The electing lane id is made opaque on every trip: with a plain `lane == 0` hipcc threads
            // this branch together with a later `if (lane == 0)` of the previous trip, lane 0 then loops
            // apart from lanes 1..63 and the readfirstlane below no longer sees lane 0 (observed: hang).
            int elect = lane;
            asm volatile("" : "+v"(elect));
            if (elect == 0) tid = atomicAdd(P.queue, 1u);
        }
        tid = __builtin_amdgcn_readfirstlane(tid);
        if (tid >= P.n_tasks) break;

        const BatchTask task = P.tasks[tid];
        const int m = (int)task.text_len;
        const uint32_t* tx = reinterpret_cast<const uint32_t*>(P.arena + task.text_off);
        const uint32_t slot = task.slot0 + lane;
        const uint32_t poff = P.slot_poff[slot];
        const int n = (int)P.slot_plen[slot];
        const uint32_t outi = P.slot_out[slot];
        // LANES: task.text_len is the longest text of the task; all columns run as full, masked blocks
        const int nblk = LANES ? (m + 3) >> 2 : m >> 2, rem = LANES ? 0 : m & 3;
        const uint32_t* txl = tx;
        int ml = m, last_dw = 0;
        if (LANES) {
            txl = reinterpret_cast<const uint32_t*>((MODE == BM_NWG ? P.lane_text : P.arena) + P.slot_toff[slot]);
            ml = (int)P.slot_tlen[slot];
            last_dw = ml > 0 ? (ml - 1) >> 2 : 0;   // loads never leave the lane's own text (+ arena slack)
        }
        const int front_pad = LANES && MODE == BM_NWG ? 4 * nblk - ml : 0;   // p: columns ahead of the lane's own text
        // the lane's text word for block jb, symbols past the text replaced by the pad symbol
        auto lane_word = [&](int jb) -> uint32_t {
            if (MODE == BM_NWG) return txl[min(jb, nblk - 1)];   // pre-padded row of 4 * nblk codes
            const uint32_t w = txl[min(jb, last_dw)];
            const int valid = ml - 4 * jb;
            const uint32_t keep = valid >= 4 ? 0xffffffffu : (valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u));
            return (w & keep) | (P.tpad_word & ~keep);
        };

        int best = 0;
        int nw_score = 0;

        for (int s = 0; s < (int)task.n_strips; ++s) {
            const int row0 = s * R;
            // ---- this strip's pattern symbols, padded past the pattern's end
            uint32_t pk[Q];
            {
                const uint32_t* pp = reinterpret_cast<const uint32_t*>(P.arena + poff + row0);
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const int valid = n - (row0 + 4 * q);   // symbols of this dword inside the pattern
                    uint32_t w = pp[q];
                    const uint32_t keep = valid >= 4 ? 0xffffffffu : (valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u));
                    pk[q] = (w & keep) | (P.pad_word & ~keep);
                }
            }
            // ---- column 0 of the strip (hw2.cpp:125-130; SW: zeros, 193)
            int H[R], Hs[MODE == BM_SWS ? R : 1];
#pragma unroll
            for (int r = 0; r < R; ++r) H[r] = (MODE == BM_NW) ? mulw(row0 + r + 1, P.gap) : 0;
#pragma unroll
            for (int r = 0; r < (MODE == BM_SWS ? R : 1); ++r) Hs[r] = 0;
            int topprev = (MODE == BM_NW) ? mulw(row0, P.gap) : 0;

            const int32_t* hin = hand + (size_t)((s + 1) & 1) * P.hand_half;   // written by strip s-1
            int32_t* hout = hand + (size_t)(s & 1) * P.hand_half;
            const bool has_top = s > 0;
            const bool has_bot = s + 1 < (int)task.n_strips;

            const size_t in_stride = has_top ? 64 : 0, out_stride = has_bot ? 64 : 0;
            const int4* hin4 = reinterpret_cast<const int4*>(hin) + lane;
            int4* hout4 = reinterpret_cast<int4*>(hout) + lane;
            int4 tnext = make_int4(0, 0, 0, 0);
            if (MULTI) tnext = hin4[0];
            uint32_t cwn = LANES ? lane_word(0) : tx[0];
            // single-strip form: row 0 is 0, but as a literal it makes hipcc split the first row's v_max3_i32 into
            // unsigned/signed v_max pairs that no longer fuse (+0.7 VALU per cell, [asm]); keep the zero opaque
            int zero = 0;
            asm volatile("" : "+v"(zero));
            for (int jb = 0; jb < nblk; ++jb) {
                const uint32_t cw = cwn;
                const int4 tcur = tnext;
                cwn = LANES ? lane_word(jb + 1) : tx[jb + 1];   // arena slack makes the over-read safe
                if (MULTI) tnext = hin4[(size_t)(jb + 1) * in_stride];
                int top[4], bot[4];
                uint32_t cs[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (LANES && SCORE == SC_PERM) {
                        cs[k] = __builtin_amdgcn_perm(cw, cw, 0x01010101u * (uint32_t)k);   // byte k in all four bytes
                    } else {
                        const uint32_t c = (cw >> (8 * k)) & 0xffu;
                        cs[k] = (SCORE == SC_PERM) ? c * 0x01010101u : c;
                    }
                }
                {
                    const int tl[4] = {tcur.x, tcur.y, tcur.z, tcur.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        int row0v = (MODE == BM_NW) ? mulw(4 * jb + k + 1, P.gap) : (MULTI ? 0 : zero);   // hw2.cpp:131-136
                        if (LANES && MODE == BM_NWG) row0v = mulw(min(4 * jb + k + 1, front_pad), -P.gap);   // G' of H[0][j] = g*max(j-p,0)
                        top[k] = has_top ? tl[k] : row0v;
                    }
                }
                dp_block<R, 4, MODE, SCORE>(H, Hs, pk, cs, top, topprev, bot, best, P);
                if (MULTI) hout4[(size_t)jb * out_stride] = make_int4(bot[0], bot[1], bot[2], bot[3]);
            }
            if (rem > 0) {
                uint32_t cw = cwn;
                int t0 = tnext.x, t1 = tnext.y, t2 = tnext.z;
#pragma unroll 1
                for (int k = 0; k < rem; ++k) {
                    const uint32_t c = cw & 0xffu;
                    cw >>= 8;
                    const uint32_t cs1[1] = {(SCORE == SC_PERM) ? c * 0x01010101u : c};
                    const int top1[1] = {has_top ? t0 : ((MODE == BM_NW) ? mulw(4 * nblk + k + 1, P.gap) : (MULTI ? 0 : zero))};
                    t0 = t1;
                    t1 = t2;
                    int bot1[1];
                    dp_block<R, 1, MODE, SCORE>(H, Hs, pk, cs1, top1, topprev, bot1, best, P);
                    if (MULTI) hout[((size_t)nblk * out_stride + lane) * 4 + k] = bot1[0];
                }
            }
            // ---- NW: dp[n][m] sits in this strip for the lanes whose pattern ends here (hw2.cpp:186)
            if (MODE != BM_SW && MODE != BM_SWS) {
                const int rl = n - 1 - row0;
                if (rl >= 0 && rl < R) {
                    int v = 0;
#pragma unroll
                    for (int r = 0; r < R; ++r) v = (rl == r) ? H[r] : v;
                    nw_score = v;
                }
            }
        }
        if (outi != 0xffffffffu) {
            int sc = best;
            if (MODE == BM_NW) sc = nw_score;
            if (MODE == BM_NWG) sc = addw(nw_score, mulw(n + m, P.gap));
            P.scores[outi] = sc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Two waves per task: wave 0 runs strip 0, wave 1 runs strip 1 of the same 64 pairs, CONCURRENTLY, and the
// strip boundary row goes from one to the other through a ring in LDS (16 blocks of 4 columns, 16 KiB)
// instead of through HBM.  Used when every task of a launch has at most two strips (the saturating SW
// form on 150-row patterns: 2 x 76 rows).  Same cells, same instruction count, no hand-off traffic.
#ifndef PWA_PAIR_RING
#define PWA_PAIR_RING 16
#endif
#ifndef PWA_PAIR_SLEEP
#define PWA_PAIR_SLEEP 1
#endif
#ifndef PWA_PAIR_LAG
#define PWA_PAIR_LAG 1
#endif
constexpr int kPairRing = PWA_PAIR_RING;

template <int R, int MODE, int SCORE, int ROLE>
__device__ __forceinline__ void pair_strip(const BatchParams& P, const BatchTask& task, int lane, uint32_t poff, int n, int m,
                                           const uint32_t* tx, uint32_t outi, bool has_bot, int4 (*ring)[64], uint32_t* ready,
                                           uint32_t* taken, int& best, bool& failed) {
    constexpr int Q = R / 4;
    constexpr bool TOP = ROLE == 1;
    const uint32_t spin_limit = 1u << 26;
    const int nblk = m >> 2, rem = m & 3;
    const int row0 = ROLE * R;
    uint32_t pk[Q];
    {
        const uint32_t* pp = reinterpret_cast<const uint32_t*>(P.arena + poff + row0);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int valid = n - (row0 + 4 * q);
            const uint32_t wd = pp[q];
            const uint32_t keep = valid >= 4 ? 0xffffffffu : (valid <= 0 ? 0u : ((1u << (8 * valid)) - 1u));
            pk[q] = (wd & keep) | (P.pad_word & ~keep);
        }
    }
    int H[R], Hs[MODE == BM_SWS ? R : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) H[r] = (MODE == BM_NW) ? mulw(row0 + r + 1, P.gap) : 0;
#pragma unroll
    for (int r = 0; r < (MODE == BM_SWS ? R : 1); ++r) Hs[r] = 0;
    int topprev = (MODE == BM_NW) ? mulw(row0, P.gap) : 0;
    uint32_t cwn = tx[0];
    // wave 0 keeps publishing even for a single-strip task (nobody reads): avoids a branch in the loop; the
    // ring-full wait is skipped by pretending everything was taken
    const uint32_t never_full = has_bot ? 0u : 0x40000000u;
    // row 0 of a local alignment is 0, but hipcc must not know: with a literal 0 it rewrites the first row's signed
    // v_max3_i32 into unsigned/signed v_max pairs that no longer fuse (4.7 instead of 4.0 VALU per cell, [asm])
    int zero = 0;
    asm volatile("" : "+v"(zero));
    uint32_t ready_seen = 0, taken_seen = never_full;   // what this wave last saw of the other wave's progress
    for (int jb = 0; jb < nblk; ++jb) {
        const uint32_t cw = cwn;
        cwn = tx[jb + 1];   // arena slack makes the over-read safe
        int top[4], bot[4];
        uint32_t cs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t c = (cw >> (8 * k)) & 0xffu;
            cs[k] = (SCORE == SC_PERM) ? c * 0x01010101u : c;
        }
        if (TOP) {   // the row above comes from wave 0 through the LDS ring
            if (ready_seen <= (uint32_t)jb) {   // the flag is re-read only when the cached value runs out
                const uint32_t want = min((uint32_t)jb + PWA_PAIR_LAG, (uint32_t)nblk + (rem > 0));
                for (uint32_t spins = 0; !failed && (ready_seen = __builtin_amdgcn_readfirstlane(lds_peek(ready))) < want;) {
                    __builtin_amdgcn_s_sleep(PWA_PAIR_SLEEP);
                    if (++spins > spin_limit) failed = true;
                }
            }
            const int4 tcur = ring[jb % kPairRing][lane];
            lds_post(taken, (uint32_t)jb + 1);
            top[0] = tcur.x; top[1] = tcur.y; top[2] = tcur.z; top[3] = tcur.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) top[k] = (MODE == BM_NW) ? mulw(4 * jb + k + 1, P.gap) : zero;
        }
        dp_block<R, 4, MODE, SCORE>(H, Hs, pk, cs, top, topprev, bot, best, P);
        if (!TOP) {   // hand the strip's bottom row to wave 1
            if ((int)((uint32_t)jb - taken_seen) >= kPairRing) {
                for (uint32_t spins = 0; !failed && (int)((uint32_t)jb - (taken_seen = __builtin_amdgcn_readfirstlane(lds_peek(taken)) + never_full)) >= kPairRing;) {
                    __builtin_amdgcn_s_sleep(PWA_PAIR_SLEEP);
                    if (++spins > spin_limit) failed = true;
                }
            }
            ring[jb % kPairRing][lane] = make_int4(bot[0], bot[1], bot[2], bot[3]);
            lds_post(ready, (uint32_t)jb + 1);
        }
    }
    if (rem > 0) {   // the last 1..3 columns
        uint32_t c4 = cwn;
        int t0 = 0, t1 = 0, t2 = 0;
        if (TOP) {
            for (uint32_t spins = 0; !failed && lds_peek(ready) <= (uint32_t)nblk;) {
                __builtin_amdgcn_s_sleep(PWA_PAIR_SLEEP);
                if (++spins > spin_limit) failed = true;
            }
            const int4 tcur = ring[nblk % kPairRing][lane];
            lds_post(taken, (uint32_t)nblk + 1);
            t0 = tcur.x; t1 = tcur.y; t2 = tcur.z;
        }
        int b0 = 0, b1 = 0, b2 = 0;
#pragma unroll 1
        for (int k = 0; k < rem; ++k) {
            const uint32_t c = c4 & 0xffu;
            c4 >>= 8;
            const uint32_t cs1[1] = {(SCORE == SC_PERM) ? c * 0x01010101u : c};
            const int top1[1] = {TOP ? t0 : ((MODE == BM_NW) ? mulw(4 * nblk + k + 1, P.gap) : zero)};
            t0 = t1;
            t1 = t2;
            int bot1[1];
            dp_block<R, 1, MODE, SCORE>(H, Hs, pk, cs1, top1, topprev, bot1, best, P);
            b0 = (k == 0) ? bot1[0] : b0;
            b1 = (k == 1) ? bot1[0] : b1;
            b2 = (k == 2) ? bot1[0] : b2;
        }
        if (!TOP) {
            for (uint32_t spins = 0; !failed && (int)((uint32_t)nblk - (lds_peek(taken) + never_full)) >= kPairRing;) {
                __builtin_amdgcn_s_sleep(PWA_PAIR_SLEEP);
                if (++spins > spin_limit) failed = true;
            }
            ring[nblk % kPairRing][lane] = make_int4(b0, b1, b2, 0);
            lds_post(ready, (uint32_t)nblk + 1);
        }
    }
    if (MODE != BM_SW && MODE != BM_SWS) {
        const int rl = n - 1 - row0;
        if (rl >= 0 && rl < R) {
            int v = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) v = (rl == r) ? H[r] : v;
            if (outi != 0xffffffffu) P.scores[outi] = (MODE == BM_NWG) ? addw(v, mulw(n + m, P.gap)) : v;
        }
    }
}

template <int R, int MODE, int SCORE>
__global__ __launch_bounds__(128, strip_waves_per_simd(R, MODE)) void batch_scores_pair_kernel(const BatchParams P) {
    __shared__ int4 ring[kPairRing][64];
    __shared__ uint32_t ready, taken, task_sh;
    __shared__ int best_sh[64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    for (;;) {
        __syncthreads();   // both waves are done with the previous task's ring
        if (threadIdx.x == 0) {
            task_sh = atomicAdd(P.queue, 1u);
            ready = 0;
            taken = 0;
        }
        __syncthreads();
        const uint32_t tid = __builtin_amdgcn_readfirstlane(task_sh);
        if (tid >= P.n_tasks) break;

        const BatchTask task = P.tasks[tid];
        const int m = (int)task.text_len;
        const uint32_t* tx = reinterpret_cast<const uint32_t*>(P.arena + task.text_off);
        const uint32_t slot = task.slot0 + lane;
        const uint32_t poff = P.slot_poff[slot];
        const int n = (int)P.slot_plen[slot];
        const uint32_t outi = P.slot_out[slot];
        const bool active = (uint32_t)w < task.n_strips;
        const bool has_bot = task.n_strips > 1;
        int best = 0;
        bool failed = false;

        if (active) {
            // one body per role (ROLE 0: no row above, hands its bottom row on; ROLE 1: takes the row above): the
            // column loop must not carry role branches -- hipcc then doubles the live H registers and spills
            if (w == 0) pair_strip<R, MODE, SCORE, 0>(P, task, lane, poff, n, m, tx, outi, has_bot, ring, &ready, &taken, best, failed);
            else pair_strip<R, MODE, SCORE, 1>(P, task, lane, poff, n, m, tx, outi, false, ring, &ready, &taken, best, failed);
        }
        if (failed && lane == 0) __hip_atomic_store(P.queue + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == BM_SW || MODE == BM_SWS) {   // the pair's score is the larger of the two strips' maxima
            if (w == 0) best_sh[lane] = best;
            __syncthreads();
            const uint32_t last = task.n_strips > 1 ? 1u : 0u;
            if ((uint32_t)w == last && outi != 0xffffffffu) P.scores[outi] = (last == 1) ? max(best, best_sh[lane]) : best;
        }
    }
}

}  // namespace pwa
