// mini_fill.hip.h -- DP fill with the traceback band in HBM for MANY pairs with SHORT patterns (gfx950 / MI355X).
//
// Replaces, like pair_fill.hip.h, hw2.cpp:119-156 (NW) / 193-231 (SW) -- for the shape of hw2's own `-g` runs: a few
// thousand (pattern i, reference i) pairs of ~150-row patterns (hw2.cpp:328-338), every one of which needs its traceback
// (hw2.cpp:344).
//
// Why a third mapping.  The stripe engine gives a pair one wave: 64 lanes x RL rows, i.e. a 150-row pattern runs as a
// 256-row stripe (41 % of the lanes' work and of the band bytes are padding) behind a helper wave that only stages text.
// A lane-per-pair kernel (the strip engine's mapping) would need 5 VALU per cell instead of ~14 per useful cell -- but
// 4096 pairs are 64 waves on a chip of 1024 SIMDs, each running 150 x 10 000 cells on its own: ~20 ms.  The middle:
//   * a pair = ONE DPP row of 16 lanes, lane k owns RL consecutive rows (RL = 4 .. 16: patterns of up to 256 rows, 160 for
//     RL = 10), a wave = 4 independent pairs: 4096 pairs are 1024 waves -- one per SIMD -- and a step's fixed costs (DPP
//     moves, text, store) are spread over 4 x RL rows instead of 4;
//   * the anti-diagonal front, the keyed cells (H * 4 + priority, one v_max3 per cell), table scoring on coded symbols,
//     the gap-shifted global form and the guarded chunks are pair_fill.hip.h's; `row_shr:1` moves a value from lane k-1
//     to lane k inside each 16-lane row, `row_shl:q` with bank mask 1 puts the staged text symbol of step q into lane 0
//     of every row at once -- the same two DPP moves the stripe engine uses, now serving four pairs;
//   * nothing comes from another wave: the row above a pair is the matrix's row 0 (a constant per step), the text is
//     prefetched by the lanes themselves (one byte per lane and 16-step chunk, a chunk ahead) -- no helper wave, no LDS,
//     no flags, no hand-off rows;
//   * band: per pair [step t][16 lanes][RL code bytes] in two planes per step (BandGeo<16, RL>), written with one aligned
//     4 / 8 / 16-byte store (+ one 2 / 4-byte store) per lane and step: every 16-lane row writes 16 RL contiguous bytes.
//     160 instead of 256 band bytes per column for a 150-row pattern.  The optional int32 score band is [t][16][RL].
// Tasks (4 pairs) come off an atomic queue; the host sorts pairs by text length so that a task's pairs run about the same
// number of steps, pads the list to whole tasks with empty patterns, and sizes every band for its task's longest text.
#pragma once
#include "pair_fill.hip.h"

namespace pwa {

// band steps of a mini-stripe pair whose task's longest text has m columns: m + 15 anti-diagonal steps in whole 16-step chunks
__host__ __device__ inline size_t mini_band_steps(size_t m) { return (m + 15 + 15) & ~(size_t)15; }

#ifdef PWA_BAND_STORE_NT   // experiment builds: non-temporal band stores
#define PWA_BAND_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define PWA_BAND_STORE(ptr, val) (*(ptr) = (val))
#endif
typedef uint32_t mu32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t mu32x4 __attribute__((ext_vector_type(4)));

// LN = 16: lanes 1..15 of every row <- lane k-1 of v, lane 0 of every row keeps dst (row_shr:1);  LN = 64 (one pair per wave): lanes
// 1..63 <- lane k-1, lane 0 keeps dst (wave_shr:1)
template <int LN = 16>
__device__ __forceinline__ int mini_row_shr1(int dst, int v) {
    if constexpr (LN == 16) return __builtin_amdgcn_update_dpp(dst, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
    else return __builtin_amdgcn_update_dpp(dst, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
// lane 0 of every row (LN = 64: of row 0 only) <- lane Q of v in that row (lanes 4, 8, 12 are written too and overwritten by the shift
// that follows)
template <int Q, int LN = 16>
__device__ __forceinline__ int mini_pick_lane0(int dst, int v) {
    constexpr int ROWS = LN == 16 ? 0xf : 0x1;
    if constexpr (Q == 0) return __builtin_amdgcn_update_dpp(dst, v, 0xE4 /* quad_perm:[0,1,2,3] */, ROWS, 0x1, false);
    else return __builtin_amdgcn_update_dpp(dst, v, 0x100 + Q /* row_shl:Q */, ROWS, 0x1, false);
}

// 16 steps of four pairs.  GUARD: some lane of the wave is outside its matrix at some step of the chunk (the first 15
// steps, and from the shortest text's last column on): that lane's state is frozen (pair_fill.hip.h, keyed_chunk).
template <int RL, bool LOCAL, bool SBAND, bool GUARD, bool GAP0, bool BAND = true, int LN = 16>
__device__ __forceinline__ void mini_chunk(const int t0, const int k, const int m, const uint32_t (&pk)[(RL + 3) / 4], int (&hl)[RL], int& diag0,
                                           int& bottom, int& tch, const int tcv, const int top0, const int top_inc, int (&bs)[RL], int (&bj)[RL],
                                           const uint32_t tab_lo, const uint32_t tab_hi, const int cl, g_u8* const tba, g_u8* const tbb,
                                           g_i32* const sb) {
    typedef BandGeo<LN, RL> Geo;
    constexpr int NQ = (RL + 3) / 4;
    constexpr int PU = TbCode<LOCAL>::UP, PL = TbCode<LOCAL>::LEFT;
    static_assert(!GAP0 || (!LOCAL && !SBAND && PU == 0), "gap-shifted fills: global, no score band");
    static_assert(LN == 16 || (LN == 64 && (RL == 6 || RL == 8 || RL == 12 || RL == 16)), "one pair per wave: single stripes of 384 .. 1024 rows");
    const int cu = p_addw(cl, PU - PL);
    // LOCAL: the first maximum of every row (hw2.cpp:225-229).  Inside the chunk a row keeps ONE running maximum over keys H * 16 + (15 - q)
    // -- value first, then the earlier step -- built with a v_lshl_or and folded two steps at a time by a v_max3; the row's record
    // (bs: such a key, bj: the chunk it is from) takes it at the end of the chunk if its H is strictly larger.  1.75 instead of 3
    // instructions per cell (compare + two selects).  H * 16 needs |H| < 2^26 (the host checks).
    int cmax[RL], kprev[RL];
    static_for<0, 16>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const int j = t0 + q - k + 1;
        const bool act = !GUARD || (unsigned)(j - 1) < (unsigned)m;   // this lane's column is inside its pair's matrix
        // text symbol (splatted): lane 0 of each row takes the staged symbol of step q, lane k the one lane k-1 had a step ago
        const int tn = mini_row_shr1<LN>(mini_pick_lane0<q, LN>(tch, tcv), tch);
        uint32_t s4[NQ];
#pragma unroll
        for (int x = 0; x < NQ; ++x) s4[x] = __builtin_amdgcn_perm(tab_hi, tab_lo, pk[x] ^ (uint32_t)tn);
        // the row above: lane k-1's last row of the previous step; lane 0 of each row: the matrix's row 0 (hw2.cpp:131-136 / 193)
        const int up_in = mini_row_shr1<LN>(top0 + q * top_inc, bottom);
        int dg = diag0, up = up_in;
        uint32_t codes[NQ];
        int hsb[RL];
#pragma unroll
        for (int r = 0; r < RL; ++r) {
            const int kd = p_addw(dg, (int)(int8_t)(s4[r / 4] >> (8 * (r % 4))));   // hw2.cpp:142 / 208-211: diag + s, as a key
            const int kl = hl[r];
            int kk = max(max(kd, up), kl);                                          // 142-153 / 211-222: value and direction in one max
            if (LOCAL) kk = max(kk, (int)TB_STOP);
            const int base = kk & ~3;
            if (r % 4 == 0) codes[r / 4] = tb_first_code(kk);
            if (r % 4 == 1) tb_put_code<1>(codes[r / 4], kk);
            if (r % 4 == 2) tb_put_code<2>(codes[r / 4], kk);
            if (r % 4 == 3) tb_put_code<3>(codes[r / 4], kk);
            const int hn = GAP0 ? (base | PL) : p_addw(base, cl);                   // what the next column (left) and the diagonal take
            if (LOCAL) {
                int key = (int)(((unsigned)base << 2) | (unsigned)(15 - q));
                if (GUARD) key = act ? key : 0;                                     // (a lane outside its matrix records nothing)
                // two steps per update (one v_max3), pinned in program order: left to itself hipcc turns the 16 maxima of a row into a
                // tree and keeps every key of the chunk alive for it (RL = 10: 256 VGPRs + AGPR moves, one wave per SIMD)
                if (q % 2 == 0) kprev[r] = key;
                else {
                    cmax[r] = q == 1 ? max(kprev[r], key) : max(max(cmax[r], kprev[r]), key);
                    asm volatile("" : "+v"(cmax[r]));
                }
            }
            if (SBAND) hsb[r] = kk >> 2;
            dg = kl;
            up = GAP0 ? base : p_addw(base, cu);                                    // what the row below / the lane below takes
            hl[r] = act ? hn : kl;
        }
        const int d0 = p_addw(up_in, PL - PU);                                      // dp[i_first - 1][j] in the left form: next step's diagonal of row 0
        diag0 = act ? d0 : diag0;
        bottom = act ? up : bottom;
        tch = tn;
        // the step's codes: plane A (PA bytes per lane), plane B (PB bytes per lane)
#ifdef PWA_MINI_NOSTORE   // timing-only experiment builds (results are wrong): what the fill costs without its band stores
        if (t0 == 0x7ffffff0)
#endif
        if constexpr (BAND) {
        if constexpr (Geo::PA == 4) PWA_BAND_STORE((g_u32*)(tba + q * Geo::SR), codes[0]);
        if constexpr (Geo::PA == 8) PWA_BAND_STORE((PWA_GLOBAL mu32x2*)(tba + q * Geo::SR), (mu32x2{codes[0], codes[1]}));
        if constexpr (Geo::PA == 16) PWA_BAND_STORE((PWA_GLOBAL mu32x4*)(tba + q * Geo::SR), (mu32x4{codes[0], codes[1], codes[2], codes[3]}));
        if constexpr (Geo::PB == 2) PWA_BAND_STORE((PWA_GLOBAL uint16_t*)(tbb + q * Geo::SR), (uint16_t)codes[Geo::PA / 4]);
        if constexpr (Geo::PB == 4) PWA_BAND_STORE((g_u32*)(tbb + q * Geo::SR), codes[Geo::PA / 4]);
        if (SBAND) {   // (sb: the pair's int32 band at step t0; rows in quads [quad][lane][4]: Geo::sband_off)
#pragma unroll
            for (int r = 0; r < RL; ++r) sb[q * Geo::SR + Geo::sband_off(k, r)] = hsb[r];
        }
        }
    });
    if (LOCAL) {
#pragma unroll
        for (int r = 0; r < RL; ++r) {
            const bool better = cmax[r] > (bs[r] | 15);                             // strictly larger H: the earlier chunk keeps a tie
            bs[r] = better ? cmax[r] : bs[r];
            bj[r] = better ? t0 : bj[r];
        }
    }
}

// A workgroup = FOUR waves, each running its own tasks -- four only so that the launch can be balanced: the host asks for as much (unused)
// dynamic LDS per workgroup as leaves room for exactly ceil(workgroups / CUs) of them on a CU, and the four waves of a workgroup go to
// the CU's four SIMDs.  [gpu, r03: tools/probes/simd_place2.hip] Launched as 1024 single-wave workgroups, the same kernel took 1.35 ms
// after an idle GPU or after itself, but 2.4 ms whenever another kernel (the walk, a copy kernel, any large grid) had run before it:
// the dispatcher then starts somewhere else in its round-robin and doubles waves up on some SIMDs while others stay empty -- and a
// wave that runs one 1.3 ms task cannot be rebalanced.
// BAND = false: the same fill without any band -- scores (and, local, the first-maximum end cell) of short-pattern pairs that a scores
// pass routes away from the strip engine (pwalign.hip, batch_create_impl): four pairs per wave instead of one 256-row stripe per pair.
constexpr int kMiniWaves = 4;
// LN = 64 (r03): ONE pair per wave, lane k owns RL = 6 | 8 | 12 | 16 rows -- a single stripe of 384 .. 1024 rows (BandGeo<64, RL>: for RL = 8
// and 16 the stripe engine's own layout [step][64 lanes][RL]), for batches of mid-sized patterns: one wave and 17 + 5 RL instructions per
// step where the stripe engine runs 3 - 8 pipelined stripes of 33 with their hand-offs.
template <int RL, bool LOCAL, bool SBAND, bool GAP0, bool BAND = true, int LN = 16>
__global__ __launch_bounds__(64 * kMiniWaves) void mini_fill_kernel(const PairParams G) {
    static_assert(BAND || !SBAND, "no score band without the code band");
    typedef BandGeo<LN, RL> Geo;
    constexpr int NQ = (RL + 3) / 4;
    constexpr int PPW = 64 / LN;   // pairs per wave
    constexpr int PU = TbCode<LOCAL>::UP, PL = TbCode<LOCAL>::LEFT;
    const int lane = threadIdx.x & 63, k = lane & (LN - 1), grp = lane / LN;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int match = G.match, mismatch = G.mismatch, gap = G.gap;
    // key constants (pair_fill.hip.h): diagonal (s - gap) * 4 + (prio(diag) - prio(left)), left gap * 4 + prio(left); as a byte table
    const int a_match = (int)(((unsigned)match - (unsigned)gap) * 4u + (unsigned)(TB_DIAG - PL));
    const int a_mismatch = (int)(((unsigned)mismatch - (unsigned)gap) * 4u + (unsigned)(TB_DIAG - PL));
    const int cl = (int)((unsigned)gap * 4u + (unsigned)PL);
    const uint32_t bm = (uint32_t)(uint8_t)(int8_t)a_match, bx = (uint32_t)(uint8_t)(int8_t)a_mismatch;
    const uint32_t tab_lo = bm | (bx << 8) | (bx << 16) | (bx << 24), tab_hi = bx * 0x01010101u;   // selector 0 -> match, 1..7 -> mismatch
    // Tasks are dealt statically: wave w of workgroup b runs tasks 4 b + w, 4 (b + grid) + w, ... (the host sorts them longest first; all
    // tasks of a round are about equally long, and a queue could not move a 1.3 ms task anyway).
    for (uint32_t tid = blockIdx.x * kMiniWaves + wave; tid < G.n_tasks; tid += gridDim.x * kMiniWaves) {
        // this row's pair (the host pads the descriptor list to whole tasks with empty patterns on a dump band)
        const PWA_GLOBAL PairDesc* const P = (const PWA_GLOBAL PairDesc*)(G.pairs + (size_t)tid * PPW + grp);
        const int n = P->n, m = P->m;
        g_cu8* const pat = (g_cu8*)P->pat;
        int mmax = m, mmin = m;
        if (PPW == 4) {
            mmax = max(m, __shfl_xor(m, 16));
            mmin = min(m, __shfl_xor(m, 16));
            mmax = max(mmax, __shfl_xor(mmax, 32));
            mmin = min(mmin, __shfl_xor(mmin, 32));
        }
        mmax = __builtin_amdgcn_readfirstlane(mmax);
        mmin = __builtin_amdgcn_readfirstlane(mmin);
        const int n_chunks = (mmax + (LN - 1) + 15) / 16;
        const int i_first = k * RL + 1;   // first row of this lane (1-based)
        uint32_t pk[NQ];
        int hl[RL], bs[RL], bj[RL];
#pragma unroll
        for (int x = 0; x < NQ; ++x) pk[x] = 0;
#pragma unroll
        for (int r = 0; r < 4 * NQ; ++r) {
            const int i = i_first + r;
            const uint32_t c = (r < RL && i <= n) ? (uint32_t)pat[i - 1] : 7u;   // code 7 never equals a text symbol
            pk[r / 4] |= c << (8 * (r % 4));
        }
#pragma unroll
        for (int r = 0; r < RL; ++r) {
            hl[r] = tb_stored(LOCAL || GAP0 ? 0 : p_mulw(i_first + r, gap), gap, PL);   // dp[i][0], hw2.cpp:125-130 (G: 0)
            bs[r] = 0;
            bj[r] = 0;
        }
        int diag0 = tb_stored(LOCAL || GAP0 ? 0 : p_mulw(i_first - 1, gap), gap, PL);   // dp[i_first-1][0]
        // row 0 in the form `bottom` travels in (the up-candidate of the row below): global H = j * gap, G and local 0
        const int top_inc = (LOCAL || GAP0) ? 0 : (int)((unsigned)gap * 4u);
        g_u8* const tb = (g_u8*)P->tb;
        g_i32* const sband = SBAND ? (g_i32*)P->sband : nullptr;
        const int offa = k * Geo::PA, offb = LN * Geo::PA + k * Geo::PB;
        int bottom = 0, tch = 0;
        // Text staging WITHOUT vector-memory loads: a single VMEM load in the chunk loop makes the wave wait, at every chunk, for all the
        // band stores issued before it (loads and stores share vmcnt and return in order) -- [gpu, r03] 3.8 ms instead of 1.5 ms for the
        // 4096 x (150 x 10k) batch.  The four texts are read with SCALAR loads (16 bytes per pair and chunk, a chunk ahead; lgkmcnt) and
        // each lane picks its byte: dword (lane >> 2) of the 16 loaded, byte (lane & 3) of it.
        const uint32_t* tg[PPW];
        int mg[PPW];
#pragma unroll
        for (int x = 0; x < PPW; ++x) {
            const uint64_t tp = (uint64_t)(uintptr_t)P->txt;
            const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)tp, LN * x), hi = __builtin_amdgcn_readlane((uint32_t)(tp >> 32), LN * x);
            tg[x] = (const uint32_t*)(uintptr_t)(((uint64_t)hi << 32) | lo);
            mg[x] = __builtin_amdgcn_readlane(m, LN * x);
        }
        auto stage = [&](int t0s, mu32x4 (&w)[PPW]) {   // bytes t0s .. t0s+15 of every pair's text (clamped: never more than 31 bytes past its end)
#pragma unroll
            for (int x = 0; x < PPW; ++x) {
                const int tc = min(t0s, (mg[x] + 15) & ~15);
                w[x] = *(const __attribute__((address_space(4))) mu32x4*)((uintptr_t)tg[x] + (size_t)tc);
            }
        };
        const uint32_t bsel = (uint32_t)(k & 3) * 0x01010101u;   // v_perm selector: byte (k & 3) of the dword, in all four bytes
        const int wsel = lane >> 2;   // (LN = 64: only lanes 0..15 -- the ones the per-step pick reads -- need their dword)
        mu32x4 wnext[PPW];
        stage(0, wnext);
        for (int ch = 0; ch < n_chunks; ++ch) {
            const int t0 = ch * 16;
            uint32_t wv = wnext[0][0];
#pragma unroll
            for (int x = 1; x < 4 * PPW; ++x) wv = (wsel == x) ? wnext[x >> 2][x & 3] : wv;
            const int tcv = (int)__builtin_amdgcn_perm(wv, wv, bsel);   // lane q of a row: its pair's column t0 + q, splatted (the symbol travels down the lanes that way)
            stage(t0 + 16, wnext);                                       // a chunk ahead
            const int top0 = tb_stored(LOCAL || GAP0 ? 0 : p_mulw(t0 + 1, gap), gap, PU);
            g_u8* const tbs = tb + (size_t)t0 * Geo::SR;
            g_i32* const sbs = SBAND ? sband + (size_t)t0 * Geo::SR : nullptr;
            const bool interior = t0 >= LN - 1 && t0 + 16 <= mmin;   // every lane of every pair inside its matrix
            if (interior)
                mini_chunk<RL, LOCAL, SBAND, false, GAP0, BAND, LN>(t0, k, m, pk, hl, diag0, bottom, tch, tcv, top0, top_inc, bs, bj, tab_lo, tab_hi, cl,
                                                                    tbs + offa, tbs + offb, sbs);
            else
                mini_chunk<RL, LOCAL, SBAND, true, GAP0, BAND, LN>(t0, k, m, pk, hl, diag0, bottom, tch, tcv, top0, top_inc, bs, bj, tab_lo, tab_hi, cl,
                                                                   tbs + offa, tbs + offb, sbs);
        }
        PWA_GLOBAL PairResult* const res = (PWA_GLOBAL PairResult*)P->res;
        if (!LOCAL) {
            // dp[n][m] (hw2.cpp:186): a lane's state freezes when it leaves the matrix, so the lane that holds row n still has its last
            // column's stored value H * 4 + gap * 4 + prio(left)
#pragma unroll
            for (int r = 0; r < RL; ++r)
                if (i_first + r == n) res->score = (int)((unsigned)hl[r] - (unsigned)cl) >> 2;
        } else {
            // per-lane reduction over row slots, then over the pair's 16 lanes: max score, then smallest i (hw2.cpp:225-229)
            int s_best = 0, i_best = 0, j_best = 0;   // (bs[r]: H * 16 + 15 - q, bj[r]: the chunk's first step -- mini_chunk)
#pragma unroll
            for (int r = 0; r < RL; ++r) {
                const int i = i_first + r, h = bs[r] >> 4;
                if (i <= n && h > s_best) {
                    s_best = h;
                    i_best = i;
                    j_best = bj[r] + (15 - (bs[r] & 15)) - k + 1;
                }
            }
#pragma unroll
            for (int off = LN / 2; off >= 1; off >>= 1) {
                const int so = __shfl_xor(s_best, off), io = __shfl_xor(i_best, off), jo = __shfl_xor(j_best, off);
                const bool better = so > s_best || (so == s_best && so > 0 && io < i_best);
                if (better) {
                    s_best = so;
                    i_best = io;
                    j_best = jo;
                }
            }
            if (k == 0) {
                g_i32* bp = (g_i32*)(G.best + P->first_stripe);
                bp[0] = s_best;
                bp[1] = i_best;
                bp[2] = j_best;
                bp[3] = 0;
            }
        }
    }
}

}  // namespace pwa
