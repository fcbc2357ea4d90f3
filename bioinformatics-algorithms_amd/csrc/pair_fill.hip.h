// pair_fill.hip.h -- full DP fill of a pair with the traceback band written to HBM, plus the
// traceback walk (gfx950 / MI355X).
//
// Replaces hw2.cpp:119-156 + 158-181 (NW) and hw2.cpp:193-231 + 233-257 (SW): the reference fills
// an int matrix and a char matrix row by row (5 B/cell) and walks the char matrix backwards.
//
// Mapping:
//   * the matrix is cut into horizontal STRIPES of 64*RL rows; one 64-lane wavefront (= one
//     single-wave workgroup) owns a stripe, lane k owns RL consecutive rows of it;
//   * the wave sweeps its stripe along an anti-diagonal front: at step t lane k computes column
//     c = t - k of its RL rows.  "left" and most "diag"/"up" operands are the lane's own registers;
//     the row above a lane's first row comes from lane k-1 one step earlier through a DPP
//     wave-shift (no LDS), as does the text character, which enters at lane 0 and travels down;
//   * the stripes of one pair run CONCURRENTLY as a software pipeline.  A workgroup = W compute waves
//     (W consecutive stripes, one wave per SIMD) + ONE helper wave.  Inside the workgroup stripe s+1
//     consumes the bottom row of stripe s through a column-indexed ring in LDS, ~80 steps behind it
//     (63 steps of lane skew + one 16-step chunk).  Between workgroups the row goes through HBM and
//     ONLY the helper wave touches it: it polls the producer's column counter (sc1 loads), stages
//     row + text into LDS rings, and in the other direction drains the last compute wave's output
//     ring with write-through (sc1) stores, its own vmcnt(0), and one sc1 counter store
//     (cdna_hip_programming.md Guideline 16, R1).  The compute waves therefore issue no vector-memory
//     LOADS at all in their step loop -- only the fire-and-forget band stores -- so no s_waitcnt
//     vmcnt ever sits on the DP's critical path (hipcc otherwise drains the band stores at every
//     chunk boundary as soon as one load shares the loop).  Tasks (super-stripes) are dealt in
//     global order from an atomic queue, so the producer of any task is always already running: no
//     residency assumption, no deadlock for any grid size; every spin is bounded;
//   * traceback codes (1 B/cell, the reference's char matrix) leave in SKEWED layout
//     tb[stripe][step][lane][RL]: every step each wave stores 64*RL contiguous bytes (256 B for
//     RL = 4); the optional int32 score band uses the same layout (1 KiB per step);
//   * SW keeps, per (lane, row-slot), the first strict maximum in column order; a stripe reduces
//     to (score, smallest i, its j); the walk kernel takes the first best stripe = the reference's
//     first row-major maximum (hw2.cpp:225-229).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lds_sync.hip.h"

namespace pwa {

// Traceback codes = the direction's PRIORITY in the reference's tie-break, so that they can ride in the two low bits
// of a packed key (see keyed_chunk): global (hw2.cpp:142-153) prefers diag, then left, then up; local (211-222)
// prefers the zero floor, then diag, then up, then left.
enum { TB_DIAG = 2, TB_STOP = 3 };
template <bool LOCAL> struct TbCode {
    static constexpr int UP = LOCAL ? 1 : 0, LEFT = LOCAL ? 0 : 1;
};
// Keyed form of a DP value (traceback kernels): key = H * 4 + priority.  One v_max3 over the three candidate keys picks
// the maximum AND, among equal scores, the direction the reference prefers; its low bits are the traceback code.
// A cell is kept as  stored(H) = H * 4 + (gap * 4 + prio(left)),  i.e. the candidate its RIGHT neighbour sees.
// codes byte R = key & 3, in one instruction: SDWA writes a single byte of the destination and keeps the rest
template <int R>
__device__ __forceinline__ void tb_put_code(uint32_t& codes, int key) {
    static_assert(R >= 0 && R < 4, "four rows per code dword");
    if (R == 0) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(codes) : "v"(key), "v"(3));
    if (R == 1) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(codes) : "v"(key), "v"(3));
    if (R == 2) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(codes) : "v"(key), "v"(3));
    if (R == 3) asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(codes) : "v"(key), "v"(3));
}
// ... the step's first code: byte 0 = key & 3, bytes 1..3 = 0 (no zero-initialising move in front of it)
__device__ __forceinline__ uint32_t tb_first_code(int key) {
    uint32_t codes;
    asm("v_and_b32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(codes) : "v"(key), "v"(3));
    return codes;
}
__device__ __forceinline__ int tb_stored(int h, int gap, int prio_left) { return (int)((unsigned)h * 4u + (unsigned)gap * 4u + (unsigned)prio_left); }

struct PairResult {
    int32_t score;
    uint32_t end_i, end_j;
    uint32_t start_i, start_j;
    uint32_t n_ops;
    uint32_t overflow;   // ops capacity exceeded
    int32_t overlap;     // WALK_OVERLAP: longest run of equal, gap-free columns (hw2.cpp:267-278)
};

struct StripeBest {       // per stripe task (SW)
    int32_t score;
    uint32_t i, j;
    uint32_t pad;
};

struct PairDesc {          // one pair of a launch
    const uint8_t* pat;   // n bytes (+ slack)
    const uint8_t* txt;   // m bytes (+ slack)
    int32_t n, m;
    uint8_t* tb;          // skewed traceback band (unused when the kernel is built without TB)
    int32_t* sband;       // optional skewed int32 score band (same indexing)
    int32_t* rows;        // bottom rows of super-stripes 0 .. n_super-2, row_stride int32 each
    PairResult* res;
    uint8_t* ops;         // traceback output, capacity ops_cap
    uint32_t ops_cap;
    uint32_t first_task;  // index of this pair's super-stripe 0 in the task list
    uint32_t first_stripe;// index of this pair's stripe 0 in PairParams::best
    uint32_t n_stripes;
    uint32_t row_stride;  // >= m + 64
    uint32_t out_index;   // slot of this pair in PairParams::scores_out
    int32_t score_bias;   // global fills in gap-shifted coordinates (G = H - gap (i + j), GAP0 kernels): gap * (n + m), else 0
    uint32_t pad[2];
};

struct StripeTask {       // one workgroup task: W consecutive stripes of one pair
    uint32_t pair, super;
};

struct PairParams {
    const PairDesc* pairs;
    const StripeTask* tasks;
    uint32_t n_pairs, n_tasks;
    uint32_t* queue;        // [0] atomic task counter, [1] error flag (both zeroed before every launch)
    uint32_t* progress;     // per task: bottom-row COLUMNS visible to other CUs (zeroed before every launch)
    StripeBest* best;       // per stripe (SW)
    int32_t* scores_out;    // optional device score vector in caller order (nullptr: results only in PairResult)
    int32_t match, mismatch, gap;
    int32_t trace_stripe;         // debugging: stripes trace_stripe .. +3 also record the start time of each of their first 8192 chunks
    uint32_t trace_base;          //            at stamps[trace_base + 8192 * k + chunk]
    unsigned long long* stamps;   // debugging (PWA_STAMPS): per stripe {start, first interior chunk, end, -} in s_memrealtime ticks (10 ns), or nullptr
    int32_t dash;           // WALK_OVERLAP: the arena's symbol for a literal '-' (its code when the arena is coded), or a
                            // value no symbol has when no sequence contains one (hw2.cpp:269 skips such columns)
};

__device__ __forceinline__ int p_addw(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int p_mulw(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

// lane k <- lane k-1; lane 0 keeps `fill`
__device__ __forceinline__ int wave_shr1(int fill, int v) {
    return __builtin_amdgcn_update_dpp(fill, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
// lane k <- lane k+1; lane 63 keeps `fill`
__device__ __forceinline__ int wave_shl1(int fill, int v) {
    return __builtin_amdgcn_update_dpp(fill, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

#define PWA_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Pointers that arrive inside descriptors are generic to the compiler; as FLAT accesses they would
// also tick lgkmcnt and force an lgkmcnt(0) wait into every step.  All of them are hipMalloc memory.
#define PWA_GLOBAL __attribute__((address_space(1)))
typedef PWA_GLOBAL uint8_t g_u8;
typedef PWA_GLOBAL const uint8_t g_cu8;
typedef PWA_GLOBAL uint32_t g_u32;
typedef PWA_GLOBAL int32_t g_i32;
typedef PWA_GLOBAL const int32_t g_ci32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// steps per stripe in the band layout: m + 63 anti-diagonal steps, rounded up to whole 16-step chunks (the fill always runs
// whole chunks and stores every (step, lane) slot; slots of lanes outside the matrix are padding nobody reads)
__host__ __device__ inline size_t band_steps(size_t m) { return (m + 63 + 15) & ~(size_t)15; }
template <int RL>
__device__ __forceinline__ size_t tb_index(int i, int j, int m) {   // i, j >= 1
    const int q = i - 1;
    const int s = q / (64 * RL);
    const int k = (q % (64 * RL)) / RL;
    const int r = q % RL;
    const size_t T = band_steps((size_t)m);
    return ((s * T + (size_t)(j - 1 + k)) * 64 + k) * RL + r;
}

// One anti-diagonal step of a stripe.  EDGE = some lanes of this step may lie outside the matrix
// or compute the matrix's last column.
template <int RL, bool LOCAL, bool TB, bool SBAND, bool EDGE, bool PERM = false, bool KEYED = true>
__device__ __forceinline__ void stripe_step(int t, int lane, int m, int n, int i_first, const int (&pc)[RL], int (&hl)[RL],
                                            int& diag0, int& bottom, int& tch, int& topv, int& tcv, int& coll,
                                            int (&bs)[RL], int (&bj)[RL], int match, int mismatch, int gap,
                                            g_u8* tbs, g_i32* sbs, PWA_GLOBAL PairResult* res) {
    const int up_in = wave_shr1(topv, bottom);   // dp[i_first-1][j]; lane 0: the staged row above the stripe
    tch = wave_shr1(tcv, tch);                   // text char of column c; lane 0: the staged text
    topv = wave_shl1(topv, topv);                // rotate the staged vectors: lane 0 sees the next column next step
    tcv = wave_shl1(tcv, tcv);
    const int c = t - lane;
    uint32_t codes = 0;
    int hnew[RL];
    const bool active = !EDGE || (c >= 0 && c < m);
    if (active) {
        const int j = c + 1;
        int dg = diag0, up = up_in;
        // (keyed traceback fills -- H * 4 + priority, one v_max3 per cell -- run keyed_chunk below; this step serves the plain forms)
        static_assert(!(TB && KEYED), "keyed fills run keyed_chunk");
        if (TB) {
            // plain int32 form with a band: the reference's own compare-and-select chains, for scores x lengths that leave
            // the keyed form's 2^28 range (any scoring the reference's `int` holds; one VALU chain per row, ~11 per cell)
            constexpr int PU = TbCode<LOCAL>::UP, PL = TbCode<LOCAL>::LEFT;
#pragma unroll
            for (int r = 0; r < RL; ++r) {
                const int sc = (pc[r] == tch) ? match : mismatch;
                const int tdiag = p_addw(dg, sc);
                const int lf = hl[r];
                const int ug = p_addw(up, gap), lg = p_addw(lf, gap);
                int h, code;
                if (LOCAL) {
                    h = max(0, max(tdiag, max(ug, lg)));                                                   // hw2.cpp:211
                    code = (h == 0) ? (int)TB_STOP : (h == tdiag) ? (int)TB_DIAG : (h == ug) ? PU : PL;    // 214-222
                    if (h > bs[r]) {                                                                       // 225-229
                        bs[r] = h;
                        bj[r] = j;
                    }
                } else {
                    h = tdiag;                                                                             // 142-153
                    code = TB_DIAG;
                    if (lg > h) {
                        h = lg;
                        code = PL;
                    }
                    if (ug > h) {
                        h = ug;
                        code = PU;
                    }
                    if (EDGE && (i_first + r) == n && j == m) res->score = h;                              // 186
                }
                codes |= (uint32_t)code << (8 * r);
                dg = lf;
                up = h;
                hl[r] = h;
                hnew[r] = h;
            }
        } else {
#pragma unroll
            for (int r = 0; r < RL; ++r) {
                const int sc = (pc[r] == tch) ? match : mismatch;
                const int tdiag = p_addw(dg, sc);
                const int lf = hl[r];
                const int ug = p_addw(up, gap), lg = p_addw(lf, gap);
                int h;
                if (LOCAL) {
                    h = max(0, max(tdiag, max(ug, lg)));                                                   // hw2.cpp:211
                    if (h > bs[r]) {                                                                       // 225-229
                        bs[r] = h;
                        bj[r] = j;
                    }
                } else {
                    h = max(tdiag, max(lg, ug));                                                           // 142-153
                    if (EDGE && (i_first + r) == n && j == m) res->score = h;                              // 186
                }
                dg = lf;
                up = h;
                hl[r] = h;
                hnew[r] = h;
            }
        }
        // keyed form: `bottom` (and with it the rings and hand-off rows) carries the UP-candidate form H*4 + gap*4 + prio(up),
        // what the row below feeds straight into its v_max3; diag0 stays in the left-candidate form the tables are built for
        diag0 = up_in;
        bottom = up;
    } else {
#pragma unroll
        for (int r = 0; r < RL; ++r) hnew[r] = 0;
    }
    coll = wave_shl1(bottom, coll);   // lane 63 inserts its bottom-row value (column t-63), the rest shifts down
    if (TB) {
        if (RL == 4) {
            ((g_u32*)tbs)[(size_t)t * 64 + lane] = codes;
        } else if (RL == 2) {
            ((PWA_GLOBAL uint16_t*)tbs)[(size_t)t * 64 + lane] = (uint16_t)codes;
        } else {
#pragma unroll
            for (int r = 0; r < RL; ++r) tbs[((size_t)t * 64 + lane) * RL + r] = (uint8_t)(codes >> (8 * r));
        }
    }
    if (SBAND) {
#pragma unroll
        for (int r = 0; r < RL; ++r) sbs[((size_t)t * 64 + lane) * RL + r] = hnew[r];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// kCH interior steps of the KEYED form, written out in issue order.
//
// Keyed form: a cell's candidates are H * 4 + priority keys built with fast-class adds, and ONE v_max3 replaces the two compare-and-select
// chains for the value and for the code (costs per instruction: profiles/r01_valu_class_microbench.txt).  The caller passes the key
// constants in place of the scores (wave-uniform, computed once per task): match -> (match - gap) * 4 + (prio(diag) - prio(left)),
// mismatch likewise, gap -> gap * 4 + prio(left).  PERM (sequences coded 0..6, pad 7): pc[0] holds the lane's RL codes as bytes, the text
// symbol arrives splatted, and the two constants travel as a byte table (selector 0 -> match, 1..7 -> mismatch): one v_xor + one v_perm
// per step and a sign-extending SDWA add per row instead of compare + select + add per row.
//
// A stripe is ONE wave alone on its SIMD, and the DP gives it one long dependent chain: row r's "up" candidate is row
// r-1's fresh value, and row 0's comes from the lane above through a DPP move of the last row's value of the previous
// step.  One wave alone issues an instruction every ~5.4 cycles whatever its class, and one that reads the result of the
// instruction before it waits ~4 cycles longer (profiles/r02_valu_issue_microbench.txt).  hipcc schedules the
// straightforward source as a bare chain -- it hoists the step's independent work (text / row rotations, table lookups)
// in front of it -- so a step cost ~7.4 cycles x its instruction count (r01: 275 cycles for 37 instructions).  Here
//  (1) the chain is 3 instructions per row, v_max3 -> v_and -> v_add: the up-candidate of the next row is built from
//      `base` next to, not after, the row's own stored value, and values travel between lanes / stripes in that form;
//  (2) every chain instruction is followed by independent work that fits behind it -- the NEXT row's diagonal candidate,
//      this row's code byte and stored value, the per-step chores -- and __builtin_amdgcn_sched_barrier keeps hipcc from
//      regrouping them;
//  (3) lane 0's inputs (the staged row above the stripe and the staged text) are picked out of the chunk's staging
//      registers by a row_shl:q DPP write into lane 0 of the very register the wave_shr:1 move then fills for lanes
//      1..63 (q is a compile-time constant: the chunk is fully unrolled) -- no copy, no rotation of the staged vectors;
//  (4) lane 63 writes its bottom-row value straight into the LDS ring of the stripe below (one ds_write per step, the
//      other lanes hit a dump line) instead of shifting it through a collector register.
#define PWA_SB() __builtin_amdgcn_sched_barrier(0)
constexpr int kCHsteps = 16;
template <int Q> struct StepIndex { static constexpr int value = Q; };
template <int Q, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (Q < N) {
        f(StepIndex<Q>{});
        static_for<Q + 1, N>(f);
    }
}
// dst lane 0 <- v lane Q (Q in 0..15); lanes 4, 8, 12 are written too (bank 0 of row 0) and are overwritten by the wave_shr:1
// move that follows; all other lanes keep `dst`
template <int Q>
__device__ __forceinline__ int dpp_pick_lane0(int dst, int v) {
    if constexpr (Q == 0) return __builtin_amdgcn_update_dpp(dst, v, 0xE4 /* quad_perm:[0,1,2,3] */, 0x1, 0x1, false);
    else return __builtin_amdgcn_update_dpp(dst, v, 0x100 + Q /* row_shl:Q */, 0x1, 0x1, false);
}
// lanes 1..63 <- lane k-1 of v; lane 0 keeps dst
__device__ __forceinline__ int dpp_fill_shr1(int dst, int v) { return __builtin_amdgcn_update_dpp(dst, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }

// GUARD: the chunk touches steps at which some lanes are outside the matrix (the first 63 steps of a stripe, while the lanes
// come in one per step, and the last ones, while they leave): the same code, with a lane's state frozen while it is outside
// -- one compare and RL + 2 selects per step.  (r01 ran those steps through a predicated, rolled loop at ~2.7x the cost of an
// interior step; as every stripe waits for the first ~80 steps of the stripe above, that ramp was the whole pipeline's
// start-up lag: 220 interior steps per stripe instead of 80 [gpu, tools/pair_scaling.py].)
// GAP0 (global, table scoring, no score band): the host runs the fill in gap-shifted coordinates G = H - gap (i + j), i.e. the
// same recurrence with gap 0 and scores s - 2 gap (ties and codes are unchanged: all three candidates of a cell shift by the same
// gap (i + j)); with the constants known at compile time the up-candidate IS `base` and the stored value is base | prio(left):
// 2 instead of 3 instructions behind each v_max3, 2 instead of 3 on the chain.
// BAND = false (r03): the same chunk with no band at all -- scores (and end cells) of long pairs that a scores pass keeps off the strips:
// no code bytes, no stores, one instruction per cell less.
template <int RL, bool LOCAL, bool SBAND, bool PERM, bool GUARD, bool GAP0 = false, bool BAND = true>
__device__ __forceinline__ void keyed_chunk(const int t0, const int lane, const int m, const int (&pc)[RL], int (&hl)[RL], int& diag0, int& bottom,
                                            int& tch, const int topv, const int tcv, int (&bs)[RL], int (&bj)[RL], const int tab_lo,
                                            const int tab_hi, const int cl, g_u8* tbs, g_i32* sbs, int* ring_out) {
    constexpr int PU = TbCode<LOCAL>::UP, PL = TbCode<LOCAL>::LEFT;
    static_assert(!GAP0 || (!LOCAL && PERM && !SBAND && PU == 0), "gap-shifted fills: global, table scoring, no score band");
    static_assert(BAND || !SBAND, "no score band without the code band");
    const int cu = p_addw(cl, PU - PL);
    // band pointers of this lane at step t0: the unrolled steps store at immediate offsets from them
    PWA_GLOBAL uint32_t* const tb4 = (g_u32*)tbs + (size_t)t0 * 64 + lane;
    PWA_GLOBAL uint16_t* const tb2 = (PWA_GLOBAL uint16_t*)tbs + (size_t)t0 * 64 + lane;
    g_u8* const tb1 = tbs + ((size_t)t0 * 64 + lane) * RL;
    g_i32* const sb = SBAND ? sbs + ((size_t)t0 * 64 + lane) * RL : nullptr;
    auto diag_cand = [&](int r, int dg, uint32_t sc4, int sym) -> int {   // hw2.cpp:142 / 208-211: diag + s, as a key
        return PERM ? p_addw(dg, (int)(int8_t)(sc4 >> (8 * r))) : p_addw(dg, (pc[r] == sym) ? tab_lo : tab_hi);   // non-PERM: the two key constants
    };
    // text symbol and table scores of the chunk's first step (later steps get theirs one step ahead, as fillers)
    int tn = dpp_fill_shr1(dpp_pick_lane0<0>(tch, tcv), tch);
    uint32_t s4 = PERM ? __builtin_amdgcn_perm((uint32_t)tab_hi, (uint32_t)tab_lo, (uint32_t)pc[0] ^ (uint32_t)tn) : 0u;
    int upv = dpp_pick_lane0<0>(bottom, topv);   // lane 0 of the first step's "row above"
    int tdead = tch;                             // a register whose value is dead: destination of the next text pick
    constexpr int P = 3 * RL;                    // filler slots of a step: after each of the 3 chain instructions of each row
    constexpr int F_FILL = P > 3 ? 3 : P - 1;    // slot of the wave_shr:1 half of the next text symbol: >= 3 instructions after its lane-0 pick
    // LOCAL: every row's first maximum (hw2.cpp:225-229) through ONE running maximum per chunk over keys H * 16 + (15 - q) -- value first,
    // then the earlier step; the row's record (bs: such a key, bj: the chunk's first step) takes it after the chunk if its H is strictly
    // larger (mini_fill.hip.h has the same; r03).  H * 16: the host keeps local keyed fills below 2^26.
    int cmax[RL], kprev[RL];
    static_for<0, kCHsteps>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        constexpr bool more = q + 1 < kCHsteps;
        const int j = t0 + q - lane + 1;
        const bool act = !GUARD || (unsigned)(j - 1) < (unsigned)m;   // this lane's column is inside the matrix
        // ---- chain: the row above (lane k-1's last row of the previous step; lane 0: the staged row, already in place)
        const int up_in = dpp_fill_shr1(upv, bottom);
        int kd[RL], t3[RL], hn[RL], hsb[RL];
        kd[0] = diag_cand(0, diag0, s4, tn);
        if (LOCAL) t3[0] = max(max(kd[0], hl[0]), (int)TB_STOP);   // everything but the up candidate: off the chain
        if (RL > 1) kd[1] = diag_cand(1, hl[0], s4, tn);
        PWA_SB();
        int up = up_in, tn2 = tn;
        uint32_t codes = 0, s4n = s4, xn = 0;
        bool have_x = false, have_s = false;
        auto chores = [&](int slot) {   // the next step's text symbol and table scores, spread over the filler slots
            if constexpr (more) {
                if (slot == 0) tn2 = dpp_pick_lane0<q + 1>(tdead, tcv);         // lane 0 of it, into a dead register ...
                if (slot == F_FILL) tn2 = dpp_fill_shr1(tn2, tn);                // ... lanes 1..63 of it
                if (PERM && slot == F_FILL + 1) {
                    xn = (uint32_t)pc[0] ^ (uint32_t)tn2;
                    have_x = true;
                }
                if (PERM && slot == F_FILL + 2) {
                    s4n = __builtin_amdgcn_perm((uint32_t)tab_hi, (uint32_t)tab_lo, xn);
                    have_s = true;
                }
            }
        };
#pragma unroll
        for (int r = 0; r < RL; ++r) {
            const int k = LOCAL ? max(t3[r], up) : max(max(kd[r], up), hl[r]);   // chain (hw2.cpp:142-153 / 211-222 in one v_max)
            if (LOCAL && r + 1 < RL) t3[r + 1] = max(max(kd[r + 1], hl[r + 1]), (int)TB_STOP);
            if (r + 2 < RL) kd[r + 2] = diag_cand(r + 2, hl[r + 1], s4, tn);
            chores(3 * r);
            PWA_SB();
            const int base = k & ~3;                                             // chain
            if constexpr (BAND) {
                if (r == 0) codes = tb_first_code(k);
                if (r == 1) tb_put_code<1>(codes, k);
                if (r == 2) tb_put_code<2>(codes, k);
                if (r == 3) tb_put_code<3>(codes, k);
            }
            if (r == 0) {
                const int d0 = p_addw(up_in, PL - PU);                           // chore: the next step's diagonal source of row 0
                diag0 = act ? d0 : diag0;
                asm volatile("" : "+v"(diag0));                                  // (kept as its own add: folded into the next step it keeps up_in alive)
            }
            chores(3 * r + 1);
            if (SBAND) hsb[r] = k >> 2;
            PWA_SB();
            up = GAP0 ? base : p_addw(base, cu);                                 // chain: what the row below / the lane below takes
            hn[r] = GAP0 ? (base | PL) : p_addw(base, cl);                       // what the next column (left) and the diagonal take
            chores(3 * r + 2);
            if (LOCAL) {
                int key = (int)(((unsigned)base << 2) | (unsigned)(kCHsteps - 1 - q));
                if (GUARD) key = act ? key : 0;                                  // (a lane outside the matrix records nothing)
                // two steps per update (one v_max3), pinned in program order: left to itself hipcc turns the 16 maxima of a row into a
                // tree and keeps every key of the chunk alive for it (RL = 10: 256 VGPRs + AGPR moves, one wave per SIMD)
                if (q % 2 == 0) kprev[r] = key;
                else {
                    cmax[r] = q == 1 ? max(kprev[r], key) : max(max(cmax[r], kprev[r]), key);
                    asm volatile("" : "+v"(cmax[r]));
                }
            }
            PWA_SB();
        }
        bottom = act ? up : bottom;
        if constexpr (more && PERM) {
            if (!have_x) xn = (uint32_t)pc[0] ^ (uint32_t)tn2;
            if (!have_s) s4n = __builtin_amdgcn_perm((uint32_t)tab_hi, (uint32_t)tab_lo, xn);
        }
        // lane 0 of the next step's row above, into a dead register (this step's first diagonal candidate); together with the
        // stores below it also separates the chain's last add from the DPP move that reads it (VALU write -> DPP read)
        if constexpr (more) upv = dpp_pick_lane0<q + 1>(kd[0], topv);
        ring_out[q] = bottom;             // lane 63: column t - 63 of the stripe's bottom row, into the ring of the stripe below
        if constexpr (BAND) {
            if (RL == 4) {
                tb4[q * 64] = codes;
            } else if (RL == 2) {
                tb2[q * 64] = (uint16_t)codes;
            } else {
#pragma unroll
                for (int r = 0; r < RL; ++r) tb1[q * 64 * RL + r] = (uint8_t)(codes >> (8 * r));
            }
        }
        if (SBAND) {
#pragma unroll
            for (int r = 0; r < RL; ++r) sb[q * 64 * RL + r] = hsb[r];
        }
#pragma unroll
        for (int r = 0; r < RL; ++r) hl[r] = act ? hn[r] : hl[r];
        tdead = tn;
        tn = tn2;
        s4 = s4n;
        PWA_SB();
    });
    if (LOCAL) {
        static_assert(kCHsteps == 16, "four key bits for the step inside its chunk");
#pragma unroll
        for (int r = 0; r < RL; ++r) {
            const bool better = cmax[r] > (bs[r] | 15);                          // strictly larger H: the earlier chunk keeps a tie
            bs[r] = better ? cmax[r] : bs[r];
            bj[r] = better ? t0 : bj[r];
        }
    }
    tch = tn;
}

#ifndef PWA_HELPER_NAP
#define PWA_HELPER_NAP 32   // x 64 cycles
#endif
#ifndef PWA_STEP_UNROLL
#define PWA_STEP_UNROLL 8   // [gpu] C5 fill: 4 -> 21.1 ms, 8 -> 20.1 ms, 16 -> 23.2 ms
#endif
constexpr int kCH = 16;      // steps per hand-off chunk
constexpr int kRing = 512;   // columns per LDS row ring (32 chunks)
constexpr int kTrip = 256;   // columns the helper wave moves per trip and direction (4 per lane)
// [gpu, r02] with 64 columns per trip the helper was the whole pipeline's clock: every trip costs three dependent HBM
// round trips (counter poll -> row loads; row stores -> vmcnt(0) -> counter store), ~7 us, i.e. ~115 ns = 275 cycles per
// column -- exactly the "step cost" of r01, whatever the compute waves did (RL = 2 or 4, NW or SW, 31 or 37 instructions)
// column c of a stripe's bottom row is produced by lane 63 at step t = c + 63: slot = t % kRing, so that the 16 columns of one
// hand-off chunk (t0 a multiple of 16) are 16 consecutive slots that never wrap
__device__ __forceinline__ int ring_slot(int c) { return (c + 63) & (kRing - 1); }
constexpr int kTRing = 4096; // text bytes staged in LDS

template <int W>
struct WgShared {
    int ring[W + 1][kRing];       // ring[w]: row above compute wave w; ring[W]: bottom row of the last wave
    int dump[W][64 + kCH];        // keyed interior chunks: where lanes 0..62 (and lane 63 of a stripe without a consumer) put their per-step ring write
    uint8_t text[kTRing];
    uint32_t ready[W + 1];        // columns written into ring[w]
    uint32_t taken[W + 1];        // columns consumed from ring[w]
    uint32_t txt_ready;
    uint32_t task;
};

template <int RL, int W, bool LOCAL, bool TB, bool SBAND, bool PERM = false, bool KEYED = true, bool GAP0 = false, bool BAND = true>
__global__ __launch_bounds__(64 * (W + 1)) void pair_fill_kernel(const PairParams G) {
    static_assert(!PERM || (TB && KEYED), "table scoring exists for the keyed (traceback) form only");
    static_assert(BAND || (TB && KEYED && PERM && !SBAND), "the band-less keyed form: scores / end cells of coded sequences");
    constexpr bool TBK = TB && KEYED;   // values travel as H * 4 + priority
    constexpr int CH = kCH;
    __shared__ WgShared<W> sh;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int match = G.match, mismatch = G.mismatch, gap = G.gap;
    const uint32_t spin_limit = 1u << 26;
    for (;;) {
        __syncthreads();   // everybody is done with the previous task's LDS state
        if (threadIdx.x == 0) sh.task = atomicAdd(G.queue, 1u);
        if (threadIdx.x < 2 * (W + 1) + 1) {
            if (threadIdx.x <= W) sh.ready[threadIdx.x] = 0;
            else if (threadIdx.x <= 2 * W + 1) sh.taken[threadIdx.x - (W + 1)] = 0;
            else sh.txt_ready = 0;
        }
        __syncthreads();
        const uint32_t tid = __builtin_amdgcn_readfirstlane(sh.task);
        if (tid >= G.n_tasks) break;
        const StripeTask task = G.tasks[tid];
        const PairDesc P = G.pairs[task.pair];
        const int ss = (int)task.super;
        const int n = P.n, m = P.m;
        const int T = m + 63;
        const int n_chunks = (T + CH - 1) / CH;
        const int n_super = ((int)P.n_stripes + W - 1) / W;
        const bool top_global = ss > 0, bot_global = ss + 1 < n_super;
        const int wl = min(W - 1, (int)P.n_stripes - 1 - ss * W);   // last active compute wave
        g_cu8* txt = (g_cu8*)P.txt;

        if (wave == W) {
            // =================== helper wave: every global-memory hand-off of this workgroup ===================
            g_i32* rin = (g_i32*)(P.rows + (size_t)(top_global ? ss - 1 : 0) * P.row_stride);
            g_i32* rout = (g_i32*)(P.rows + (size_t)ss * P.row_stride);
            g_u32* prog_in = (g_u32*)(G.progress + (top_global ? tid - 1 : tid));   // previous super-stripe, same pair
            g_u32* prog_out = (g_u32*)(G.progress + tid);
            int kin = 0, kout = 0;
            uint32_t idle = 0;
            const bool traced = G.stamps && G.trace_stripe >= 0 && (int)(P.first_stripe + ss * W) == G.trace_stripe;   // debugging
            uint32_t trip = 0;
            for (;;) {
                const bool done_in = kin >= m, done_out = !bot_global || kout >= m;
                if (done_in && done_out) break;
                bool progress = false;
                if (traced && lane == 0 && trip < 8192) {   // slots 2 / 3 of the trace: time and columns staged / published so far
                    G.stamps[(size_t)G.trace_base + 2 * 8192 + trip] = __builtin_amdgcn_s_memrealtime();
                    G.stamps[(size_t)G.trace_base + 3 * 8192 + trip] = ((unsigned long long)(unsigned)kin << 32) | (unsigned)kout;
                }
                ++trip;
                if (!done_in) {   // ---- stage text + the row above wave 0, up to kTrip columns per trip
                    int lim = min(m, min((int)lds_peek(&sh.taken[0]) + kRing, (int)lds_peek(&sh.taken[wl]) + kTRing));
                    if (top_global) lim = min(lim, (int)__hip_atomic_load(prog_in, PWA_RLX_AGENT));   // sc1 poll
                    const int hi = min(lim, kin + kTrip);
                    if (hi > kin) {
                        int v[kTrip / 64], tc[kTrip / 64];
#pragma unroll
                        for (int u = 0; u < kTrip / 64; ++u) {   // all loads of the trip in flight together
                            const int c = kin + u * 64 + lane;
                            v[u] = tc[u] = 0;
                            if (c < hi) {
                                if (top_global) v[u] = __hip_atomic_load(rin + c, PWA_RLX_AGENT);   // sc1: issued after the poll's value is known
                                else v[u] = LOCAL ? 0 : p_mulw(c + 1, gap);                         // dp[0][j], hw2.cpp:131-136
                                tc[u] = txt[c];
                            }
                        }
#pragma unroll
                        for (int u = 0; u < kTrip / 64; ++u) {
                            const int c = kin + u * 64 + lane;
                            if (c < hi) {
                                if (TBK && !top_global) v[u] = tb_stored(v[u], gap, TbCode<LOCAL>::UP);   // the form `bottom` travels in
                                sh.ring[0][ring_slot(c)] = v[u];
                                sh.text[c % kTRing] = (uint8_t)tc[u];
                            }
                        }
                        lds_post(&sh.ready[0], (uint32_t)hi);
                        lds_post(&sh.txt_ready, (uint32_t)hi);
                        kin = hi;
                        progress = true;
                    }
                }
                if (!done_out) {   // ---- publish the bottom row of the last wave
                    const int hi = min((int)lds_peek(&sh.ready[W]), kout + kTrip);
                    if (hi > kout) {
#pragma unroll
                        for (int u = 0; u < kTrip / 64; ++u) {
                            const int c = kout + u * 64 + lane;
                            if (c < hi) __hip_atomic_store(rout + c, sh.ring[W][ring_slot(c)], PWA_RLX_AGENT);   // sc1 (write-through)
                        }
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   // only this wave's own stores
                        if (lane == 0) __hip_atomic_store(prog_out, (uint32_t)hi, PWA_RLX_AGENT);
                        lds_post(&sh.taken[W], (uint32_t)hi);
                        kout = hi;
                        progress = true;
                    }
                }
                if (progress) {
                    idle = 0;
                } else {
                    // a helper that only stages text for its own workgroup is hundreds of columns ahead of it: long naps (a batch
                    // of single-stripe pairs has three of these spinning on every SIMD next to the waves that do the work)
                    if (top_global || bot_global) __builtin_amdgcn_s_sleep(2);
                    else __builtin_amdgcn_s_sleep(PWA_HELPER_NAP);
                    if (++idle > spin_limit) {   // bounded: flag the failure, let the host report it
                        if (lane == 0) __hip_atomic_store((g_u32*)(G.queue + 1), 1u, PWA_RLX_AGENT);
                        break;
                    }
                }
            }
        } else if (wave <= wl) {
            // =================== compute wave `wave`: stripe ss*W + wave ===================
            const int s = ss * W + wave;
            const bool has_out = wave < wl || (wave == W - 1 && bot_global);
            const int i_first = s * 64 * RL + lane * RL + 1;   // first row of this lane (1-based)
            int pc[RL], hl[RL], bs[RL], bj[RL];
#pragma unroll
            for (int r = 0; r < RL; ++r) {
                const int i = i_first + r;
                pc[r] = (i <= n) ? (int)((g_cu8*)P.pat)[i - 1] : (PERM ? 7 : 256);   // 256 / code 7 never equal a text symbol
                hl[r] = LOCAL ? 0 : p_mulw(i, gap);                     // dp[i][0], hw2.cpp:125-130
                if (TBK) hl[r] = tb_stored(hl[r], gap, TbCode<LOCAL>::LEFT);
                bs[r] = 0;
                bj[r] = 0;
            }
            int diag0 = LOCAL ? 0 : p_mulw(i_first - 1, gap);           // dp[i_first-1][0]
            if (TBK) diag0 = tb_stored(diag0, gap, TbCode<LOCAL>::LEFT);
            const size_t Tb = band_steps((size_t)m);
            g_u8* tbs = (TB && BAND) ? (g_u8*)(P.tb + (size_t)s * Tb * 64 * RL) : nullptr;
            g_i32* sbs = SBAND ? (g_i32*)(P.sband + (size_t)s * Tb * 64 * RL) : nullptr;
            PWA_GLOBAL PairResult* res = (PWA_GLOBAL PairResult*)P.res;
            // traceback kernels: keyed_chunk takes the key constants instead of the three scores
            int a_match = TBK ? (int)(((unsigned)match - (unsigned)gap) * 4u + (unsigned)(TB_DIAG - TbCode<LOCAL>::LEFT)) : match;
            int a_mismatch = TBK ? (int)(((unsigned)mismatch - (unsigned)gap) * 4u + (unsigned)(TB_DIAG - TbCode<LOCAL>::LEFT)) : mismatch;
            const int a_gap = TBK ? (int)((unsigned)gap * 4u + (unsigned)TbCode<LOCAL>::LEFT) : gap;
            if (PERM) {   // ... or, coded sequences, the byte table built from them (the host checked that both fit a byte)
                const uint32_t bm = (uint32_t)(uint8_t)(int8_t)a_match, bx = (uint32_t)(uint8_t)(int8_t)a_mismatch;
                a_match = (int)(bm | (bx << 8) | (bx << 16) | (bx << 24));   // selectors 0..3
                a_mismatch = (int)(bx * 0x01010101u);                        // selectors 4..7
#pragma unroll
                for (int r = 1; r < RL; ++r) pc[0] |= pc[r] << (8 * r);
#pragma unroll
                for (int r = RL; r < 4; ++r) pc[0] |= 7 << (8 * r);
            }
            int* rin = sh.ring[wave];
            int* rout = sh.ring[wave + 1];
            int bottom = 0, tch = 0, coll = 0;
            bool failed = false;
            // keyed fills look one chunk ahead: flags and staged values of chunk ch + 1 are read (not waited for) before chunk ch
            // runs, so that a stripe that is not waiting for its producer -- the first one sets the pace of all -- pays no LDS
            // round trips between chunks; when the flags were not there yet the chunk start falls back to the polling loop
            int p_topv = 0, p_tcv = 0;
            uint32_t p_ready = 0, p_txt = 0, p_taken = 0;
            if (G.stamps && lane == 0) G.stamps[(size_t)(P.first_stripe + s) * 4 + 0] = __builtin_amdgcn_s_memrealtime();
            for (int ch = 0; ch < n_chunks; ++ch) {
                const int t0 = ch * CH;
                if (G.stamps && lane == 0 && (ch == 1 || ch == 5)) G.stamps[(size_t)(P.first_stripe + s) * 4 + (ch == 1 ? 3 : 1)] = __builtin_amdgcn_s_memrealtime();
                if (G.stamps && lane == 0 && G.trace_stripe >= 0 && (int)(P.first_stripe + s) >= G.trace_stripe && (int)(P.first_stripe + s) < G.trace_stripe + 2 && ch < 8192)
                    G.stamps[(size_t)G.trace_base + (size_t)((int)(P.first_stripe + s) - G.trace_stripe) * 8192 + ch] = __builtin_amdgcn_s_memrealtime();
                // ---- wait for the row above and the text of columns t0 .. t0+CH-1, then take them
                const uint32_t need = (uint32_t)min(m, t0 + CH);
                const int c0 = t0 + lane;
                int topv = 0, tcv = 0;
                // the look-ahead of the previous chunk when its flags covered this chunk, else the polling loop.  (Tried in r02: re-issuing
                // the look-ahead reads, one LDS round trip per attempt, instead of the loop -- it narrows the chunk-time distribution
                // mid-chain (p90 188 -> 156 ticks) but costs the common case more than it gains: C5 13.2 -> 13.4 ms, A/B on one box.)
                if (TBK && __builtin_amdgcn_readfirstlane(p_ready) >= need && __builtin_amdgcn_readfirstlane(p_txt) >= need) {
                    topv = p_topv;
                    tcv = p_tcv;
                } else {
                    for (uint32_t spins = 0; !failed && (lds_peek(&sh.ready[wave]) < need || lds_peek(&sh.txt_ready) < need);) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > spin_limit) failed = true;
                    }
                    if (lane < CH && c0 < m) {
                        topv = rin[ring_slot(c0)];
                        tcv = sh.text[c0 % kTRing];
                        if (PERM) tcv *= 0x01010101;   // the text symbol travels down the lanes already splatted
                    }
                }
                lds_post(&sh.taken[wave], need);
                const bool interior = t0 >= 63 && t0 + CH < m;   // every lane inside the matrix, last column not touched
                if constexpr (TBK) {
                    static_assert(kCHsteps == kCH, "keyed_chunk runs one hand-off chunk");
                    // the chunk's 16 bottom-row columns t0-63 .. t0-48 go into the ring step by step: make room first (the
                    // unclamped column count: lane 63 also writes while it is outside the matrix, and those slots must be free)
                    if (has_out && (t0 - 63 + CH) - (int)__builtin_amdgcn_readfirstlane(p_taken) > kRing)   // (a stale count only errs on the safe side)
                        for (uint32_t spins = 0; !failed && (t0 - 63 + CH) - (int)lds_peek(&sh.taken[wave + 1]) > kRing;) {   // ring full
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > spin_limit) failed = true;
                        }
                    if (ch + 1 < n_chunks) {   // look ahead: chunk ch + 1 (flags first, then the values they cover: LDS runs in order)
                        p_ready = lds_peek(&sh.ready[wave]);
                        p_txt = lds_peek(&sh.txt_ready);
                        if (has_out) p_taken = lds_peek(&sh.taken[wave + 1]);
                        const int c1 = c0 + CH;
                        p_topv = p_tcv = 0;
                        if (lane < CH && c1 < m) {
                            p_topv = rin[ring_slot(c1)];
                            p_tcv = sh.text[c1 % kTRing];
                            if (PERM) p_tcv *= 0x01010101;
                        }
                    }
                    int* const ring_out = (has_out && lane == 63) ? rout + ring_slot(t0 - 63) : sh.dump[wave] + lane;
                    if (interior)
                        keyed_chunk<RL, LOCAL, SBAND, PERM, false, GAP0, BAND>(t0, lane, m, pc, hl, diag0, bottom, tch, topv, tcv, bs, bj, a_match, a_mismatch,
                                                                   a_gap, tbs, sbs, ring_out);
                    else
                        keyed_chunk<RL, LOCAL, SBAND, PERM, true, GAP0, BAND>(t0, lane, m, pc, hl, diag0, bottom, tch, topv, tcv, bs, bj, a_match, a_mismatch,
                                                                  a_gap, tbs, sbs, ring_out);
                    const int hi = min(m, t0 - 63 + CH);
                    if (has_out && hi > 0) lds_post_after_writes(&sh.ready[wave + 1], (uint32_t)hi);   // after the chunk's ring writes (one wave: in order)
                    continue;
                } else {
                    if (interior) {
#pragma unroll PWA_STEP_UNROLL
                        for (int q = 0; q < CH; ++q)
                            stripe_step<RL, LOCAL, TB, SBAND, false, PERM, KEYED>(t0 + q, lane, m, n, i_first, pc, hl, diag0, bottom, tch, topv,
                                                                     tcv, coll, bs, bj, a_match, a_mismatch, a_gap, tbs, sbs, res);
                    } else {
                        const int qn = min(CH, T - t0);
#pragma unroll 1
                        for (int q = 0; q < qn; ++q)
                            stripe_step<RL, LOCAL, TB, SBAND, true, PERM, KEYED>(t0 + q, lane, m, n, i_first, pc, hl, diag0, bottom, tch, topv,
                                                                    tcv, coll, bs, bj, a_match, a_mismatch, a_gap, tbs, sbs, res);
#pragma unroll 1
                        for (int q = qn; q < CH; ++q) coll = wave_shl1(bottom, coll);   // keep the collector aligned
                    }
                }
                // ---- bottom row out: after the chunk lane 64-CH+q holds column t0 - 63 + q
                if (has_out) {
                    const int hi = min(m, t0 - 63 + CH);
                    if (hi > 0) {
                        for (uint32_t spins = 0; !failed && hi - (int)lds_peek(&sh.taken[wave + 1]) > kRing;) {   // ring full
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > spin_limit) failed = true;
                        }
                        const int c = t0 - 63 + (lane - (64 - CH));
                        if (lane >= 64 - CH && c >= 0 && c < m) rout[ring_slot(c)] = coll;
                        lds_post(&sh.ready[wave + 1], (uint32_t)hi);
                    }
                }
            }
            if (failed && lane == 0) __hip_atomic_store((g_u32*)(G.queue + 1), 1u, PWA_RLX_AGENT);
            if (G.stamps && lane == 0) G.stamps[(size_t)(P.first_stripe + s) * 4 + 2] = __builtin_amdgcn_s_memrealtime();
            if (TBK && !LOCAL) {
                // dp[n][m] (hw2.cpp:186): a lane's state freezes when it leaves the matrix, so the row that holds row n still has
                // its last column's stored value: H * 4 + gap * 4 + prio(left)
#pragma unroll
                for (int r = 0; r < RL; ++r)
                    if (i_first + r == n) res->score = (int)((unsigned)hl[r] - (unsigned)a_gap) >> 2;
            }

            if (LOCAL) {
                // per-lane reduction over row slots, then over the wave: max score, then smallest i
                int s_best = 0, i_best = 0, j_best = 0;
#pragma unroll
                for (int r = 0; r < RL; ++r) {
                    const int i = i_first + r;
                    // keyed fills: bs[r] = H * 16 + 15 - q, bj[r] = the chunk's first step (keyed_chunk); plain fills: H and j
                    const int h = TBK ? (bs[r] >> 4) : bs[r];
                    if (i <= n && h > s_best) {   // slots in increasing i: strict '>' keeps the smallest i
                        s_best = h;
                        i_best = i;
                        j_best = TBK ? bj[r] + (15 - (bs[r] & 15)) - lane + 1 : bj[r];
                    }
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const int so = __shfl_xor(s_best, off), io = __shfl_xor(i_best, off), jo = __shfl_xor(j_best, off);
                    const bool better = so > s_best || (so == s_best && so > 0 && io < i_best);
                    if (better) { s_best = so; i_best = io; j_best = jo; }
                }
                if (lane == 0) {
                    g_i32* bp = (g_i32*)(G.best + P.first_stripe + s);
                    bp[0] = s_best;
                    bp[1] = i_best;
                    bp[2] = j_best;
                    bp[3] = 0;
                }
            }
        }
    }
}

// Picks the walk's start cell, then walks hw2.cpp:163-181 (NW) / 239-257 (SW) over the skewed band.
// One wavefront per pair.  The walk is a dependent chain of n+m one-byte reads, but it only ever
// moves BACKWARDS through the band, at most two steps (512 B) per op and always inside one stripe
// until it crosses into the stripe above.  So the band is consumed in windows of WIN consecutive
// steps (WIN x 256 B, contiguous in HBM thanks to the skewed layout), staged into LDS by LDS-DMA
// (global_load_lds, 1 KiB per instruction, no VGPR round trip); while the walk runs inside one
// window the window before it is already in flight into the second LDS buffer.  The walk state
// is wave-uniform and lives in SGPRs; the only per-op latency left is one ds_read_u8.
enum { WALK_NONE = 0, WALK_OPS = 1, WALK_OVERLAP = 2 };   // end cells only / op list / overlap length, no op list
#ifndef PWA_WALK_LOAD_AUX
#define PWA_WALK_LOAD_AUX 0   // cache-policy bits of the walk's band loads (gfx94x: sc0 = 1, nt = 2, sc1 = 16)
#endif

// Where the code of a cell sits inside its band, for the two band geometries:
//   LN = 64  the stripe engine above: a stripe = 64 lanes x RL rows, one step = 64 RL bytes [lane][RL];
//   LN = 16  the mini-stripe engine (mini_fill.hip.h): a pair = ONE stripe of 16 lanes x RL rows, one step = 16 RL bytes in two
//            planes [16 lanes][PA bytes][16 lanes][PB bytes] (PA + PB = RL: what one aligned 4 / 8 / 16-byte store and one
//            2 / 4-byte store of a lane hold).
// Cell (i, j): q = i - 1, stripe q / SR, row in stripe ql = q % SR, lane k = ql / RL, band step t = j - 1 + k.
template <int LN, int RL>
struct BandGeo {
    static constexpr int SR = LN * RL;   // rows per stripe = bytes per band step
    static constexpr int PA = RL < 4 ? RL : (RL >= 16 ? 16 : (RL >= 8 ? 8 : 4)), PB = RL - PA;   // (RL = 2, 4, 8, 16: one plane, [lane][RL])
    static_assert((LN == 64 || LN == 16) && RL >= 2 && RL <= 16 && (PB == 0 || PB == 2 || PB == 4), "band geometry");
    __host__ __device__ static inline int stripe(unsigned q) { return (int)(q / (unsigned)SR); }
    __host__ __device__ static inline int row_in_stripe(unsigned q) { return (int)(q % (unsigned)SR); }
    __host__ __device__ static inline int lane(int ql) { return (int)((unsigned)ql / (unsigned)RL); }
    __host__ __device__ static inline int off(int ql) {   // byte of the cell inside its step
        if (PB == 0) return ql;
        const int k = (int)((unsigned)ql / (unsigned)RL), r = ql - k * RL;
        return r < PA ? k * PA + r : LN * PA + k * PB + (r - PA);
    }
    // int32 score band of the mini-stripe kernels, per step: quads of rows [RL / 4][LN lanes][4] and, for RL % 4 = 2, a tail [LN][2] --
    // every store instruction of a wave then writes ONE contiguous run (16 bytes per lane), whatever RL
    __host__ __device__ static inline int sband_off(int k, int r) {
        return r < (RL & ~3) ? (r >> 2) * (LN * 4) + k * 4 + (r & 3) : (RL & ~3) * LN + k * (RL & 3) + (r & 3);
    }
};

#ifndef PWA_WALK_NARROW
#define PWA_WALK_NARROW 1   // (0: experiment builds that fetch whole steps)
#endif
#ifndef PWA_WALK_WIN_WIDE_DIV
#define PWA_WALK_WIN_WIDE_DIV 1   // (experiment builds: 2 halves the window of the one-pair-per-wave classes)
#endif
template <int RL, bool LOCAL, int WALK, int LN = 64>
__global__ __launch_bounds__(64) void pair_traceback_kernel(const PairParams G) {
    typedef BandGeo<LN, RL> Geo;
    // steps per LDS window.  Stripe engine (LN = 64, RL = 2 | 4: few, long walks): 64 steps, <= 16 KiB.  Mini-stripe classes (many short
    // walks, LDS decides how many run per CU): 32 steps -- [gpu, r03] 65 536 walks over 150 x 2000 bands 4.7 -> 3.05 ms, the `g` batch
    // 0.362 -> 0.277 ms; 64 walks over 10k x 10k bands prefer the long window (0.48 against 0.585 ms)
    constexpr int WIN = LN == 16 ? 32 : (LN * RL <= 256 ? 64 : (LN * RL <= 512 ? 32 : 16) / PWA_WALK_WIN_WIDE_DIV);
    static_assert((WIN & (WIN - 1)) == 0 && WIN * LN * RL <= 16384 && (WIN * LN * RL) % 1024 == 0, "window steps");
    constexpr int STEP_BYTES = LN * RL;
    constexpr int WB = WIN * STEP_BYTES;           // bytes per window (16 KiB for RL = 4)
    __shared__ __attribute__((aligned(16))) uint8_t win[WALK != WALK_NONE ? 2 * WB + 16 : 16];   // + a byte that reads "no code"
    const int lane = threadIdx.x;
    const uint32_t pid = blockIdx.x;
    if (pid >= G.n_pairs) return;
    const PairDesc P = G.pairs[pid];
    const int m = P.m;
    const size_t T = band_steps((size_t)m);
    g_cu8* tb = (g_cu8*)P.tb;
    g_u8* ops = (g_u8*)P.ops;
    PWA_GLOBAL PairResult* res = (PWA_GLOBAL PairResult*)P.res;
    int i, j;
    if (LOCAL) {
        int sb = 0;
        i = 0;
        j = 0;
        for (uint32_t s = 0; s < P.n_stripes; ++s) {   // stripes in increasing i: strict '>' = first row-major maximum
            const PWA_GLOBAL StripeBest* b = (const PWA_GLOBAL StripeBest*)(G.best + P.first_stripe + s);
            const int sc = b->score;
            if (sc > sb) {
                sb = sc;
                i = (int)b->i;
                j = (int)b->j;
            }
        }
        if (lane == 0) res->score = sb;
        if (lane == 0 && G.scores_out) ((g_i32*)G.scores_out)[P.out_index] = sb;
    } else {
        i = P.n;
        j = P.m;
        if (lane == 0) {   // dp[n][m], written by the fill (hw2.cpp:186) -- in its own coordinates
            const int sc = p_addw(res->score, P.score_bias);
            res->score = sc;
            if (G.scores_out) ((g_i32*)G.scores_out)[P.out_index] = sc;
        }
    }
    i = __builtin_amdgcn_readfirstlane(i);
    j = __builtin_amdgcn_readfirstlane(j);
    if (lane == 0) {
        res->end_i = (uint32_t)i;
        res->end_j = (uint32_t)j;
    }
    if (WALK == WALK_NONE) return;
    // LDS-DMA of window w of stripe s into buffer `buf`: 1 KiB per instruction.  A window may run past
    // the end of its stripe or of the band; the host pads the band allocation by one window.
    // NARROW (the one-pair-per-wave classes, LN = 64 and RL >= 6: a band step is 384 .. 1024 bytes wide): only the band lanes lo .. hi of
    // every step are fetched -- a DMA lane whose 16 bytes hold none of them fetches a dummy line instead, the others land where they
    // always did, so the LDS layout does not change.  The walk only ever needs the lanes at and a few below its anchor's (rows only decrease); without the
    // mask a walk reads its pair's whole band again, and a batch of square-ish pairs spent as long in its walks as in its fills
    // ([gpu, r03] 8192 pairs 1000 x 1000: fills 1.7 ms, walks 1.65 ms = 9 GB at 5.4 TB/s).
    constexpr bool NARROW = LN == 64 && RL >= 6 && PWA_WALK_NARROW;
    constexpr int DW = RL >= 12 ? 4 : 6;   // band lanes below the anchor's that a staged window holds (the prefetched one: twice that)
    auto issue = [&](int buf, int s, int w, int lo, int hi) {
        buf = __builtin_amdgcn_readfirstlane(buf);   // (wave-uniform by construction; the DMA's LDS base travels in M0)
        const size_t off0 = ((size_t)s * T + (size_t)w * WIN) * STEP_BYTES;
        auto holds = [&](int byte_off) -> bool {   // do the 16 bytes at byte_off of the window hold a lane of lo .. hi?
            if constexpr (!NARROW) return true;
            const int bs = (int)((unsigned)byte_off % (unsigned)STEP_BYTES);
            int l0, l1;
            if (bs < LN * Geo::PA) {
                l0 = bs / Geo::PA;
                l1 = (bs + 15) / Geo::PA;
            } else {
                constexpr int PBd = Geo::PB ? Geo::PB : 1;
                l0 = (bs - LN * Geo::PA) / PBd;
                l1 = (bs - LN * Geo::PA + 15) / PBd;
            }
            return l1 >= lo && l0 <= hi;
        };
        if constexpr (NARROW) {
            // (no branch around the DMA -- its LDS base travels in M0: a masked-out lane loads the first 16 bytes of its instruction's KiB
            // instead, one cached line for all of them, into its own LDS slot, which no read touches: every read checks the lane against
            // klo.)  Which lanes of an instruction hold lanes lo .. hi repeats every NPH instructions: NPH KiB are whole steps.
            constexpr int NPH = (1024 % STEP_BYTES == 0 || STEP_BYTES % 1024 == 0) ? 1 : 3;
            static_assert((NPH * 1024) % STEP_BYTES == 0 || STEP_BYTES % 1024 == 0, "the mask pattern's period");
            uint32_t voff[NPH];
#pragma unroll
            for (int ph = 0; ph < NPH; ++ph) voff[ph] = holds(ph * 1024 + lane * 16) ? (uint32_t)lane * 16u : 0u;
#pragma unroll
            for (int u = 0; u < WB / 1024; ++u) {
                const PWA_GLOBAL uint32_t* g = (const PWA_GLOBAL uint32_t*)(tb + off0 + (size_t)(u / 4) * 4096 + voff[u % NPH]);
                __attribute__((address_space(3))) uint32_t* l = (__attribute__((address_space(3))) uint32_t*)(win + buf * WB + (u / 4) * 4096);
                if (u % 4 == 0) __builtin_amdgcn_global_load_lds(g, l, 16, 0, PWA_WALK_LOAD_AUX);
                if (u % 4 == 1) __builtin_amdgcn_global_load_lds(g, l, 16, 1024, PWA_WALK_LOAD_AUX);
                if (u % 4 == 2) __builtin_amdgcn_global_load_lds(g, l, 16, 2048, PWA_WALK_LOAD_AUX);
                if (u % 4 == 3) __builtin_amdgcn_global_load_lds(g, l, 16, 3072, PWA_WALK_LOAD_AUX);
            }
            return;
        }
        // (four pieces per address pair: the instruction's immediate offset moves the global and the LDS address alike)
#pragma unroll
        for (int u = 0; u < WB / 4096; ++u) {
            const PWA_GLOBAL uint32_t* g = (const PWA_GLOBAL uint32_t*)(tb + off0 + (size_t)u * 4096 + lane * 16);
            __attribute__((address_space(3))) uint32_t* l = (__attribute__((address_space(3))) uint32_t*)(win + buf * WB + u * 4096);
            __builtin_amdgcn_global_load_lds(g, l, 16, 0, PWA_WALK_LOAD_AUX);
            __builtin_amdgcn_global_load_lds(g, l, 16, 1024, PWA_WALK_LOAD_AUX);
            __builtin_amdgcn_global_load_lds(g, l, 16, 2048, PWA_WALK_LOAD_AUX);
            __builtin_amdgcn_global_load_lds(g, l, 16, 3072, PWA_WALK_LOAD_AUX);
        }
#pragma unroll
        for (int u = (WB / 4096) * 4; u < WB / 1024; ++u)   // (RL = 2: 8 KiB windows are whole multiples of 4 KiB as well; nothing left)
            __builtin_amdgcn_global_load_lds((const PWA_GLOBAL uint32_t*)(tb + off0 + (size_t)u * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) uint32_t*)(win + buf * WB + u * 1024), 16, 0, PWA_WALK_LOAD_AUX);
    };
    int klo0 = 0, klo1 = 0;   // NARROW: the lowest band lane staged in LDS buffer 0 / 1 (wave-uniform)
    auto klo_of = [&](int buf) { return buf ? klo1 : klo0; };
    auto set_klo = [&](int buf, int v) {
        v = __builtin_amdgcn_readfirstlane(v);
        klo0 = buf ? klo0 : v;
        klo1 = buf ? v : klo1;
    };
    // ---- the walk.  Per trip the 64 lanes look at the 64 cells of the DIAGONAL through (i, j):
    // lane d reads the code of (i-d, j-d).  The leading run of 'd' codes is one run of 'M' ops,
    // emitted by the lanes in parallel; the first non-'d' code behind it is handled in the same
    // trip.  (Typical alignments are mostly diagonal moves, so a trip retires several ops for the
    // price of one LDS round trip.)
    // WALK_OVERLAP additionally looks at the symbols under each run of diagonal moves and keeps the longest run
    // of equal ones (hw2.cpp:267-278: a gap column or a mismatch ends a run); the op list is not written.
    constexpr bool OPS = WALK == WALK_OPS, OVL = WALK == WALK_OVERLAP;
    g_cu8* pat = (g_cu8*)P.pat;
    g_cu8* txt = (g_cu8*)P.txt;
    int run = 0, best_run = 0;   // wave-uniform
    uint32_t cnt = 0;
    int cb = 0, cur_s = -1, cur_w = -1, pre_s = -1, pre_w = -1;
    bool stopped = false, walk_fault = false;
    if constexpr (OPS) {
        // ---- the op-list walk (r02).  Per trip the lanes look at 2A+1 = SEVEN diagonals of the band in one LDS round trip: lane d
        // reads
        //   view a = 0:       (i-d, j-d)        the diagonal through the anchor (i, j),
        //   view a = -1..-A:  (i+a-d, j-d)      the ones the path is on after |a| net 'u',
        //   view a = +1..+A:  (i-d, j-a-d)      ... after a net 'l',
        // each view becomes three ballot masks ('d' / gap / 'l') and a vector of op bytes, and the walk then hops between the views on
        // SCALAR bit tests alone, written out by hand (hipcc turns the state machine into a loop over state flags, ~32
        // instructions per hop).  On view a the walk stands on lane p: the run of 'd' codes from p on ends at lane q; lanes
        // p .. q (q included when it holds 'u' / 'l') store their op bytes to ops[cnt ..] (hw2.cpp:164-179 / 240-255) under
        // exec = s_bfm(count, p); 'u' moves to view a-1 (same lane while a <= 0, else lane q+1), 'l' to view a+1 (same lane
        // while a >= 0, else lane q+1); anything else -- a cell outside the staged diagonals, the staged windows or the matrix, a
        // local alignment's zero cell, lane 63 -- ends the trip there.  17 instructions per hop against ~130 per trip.
        // A lone wave issues one instruction per ~5.4 cycles whatever its kind, so the walk is priced in instructions per op:
        // the one-diagonal loop below (kept for the overlap walk) spends ~70 per trip = per non-diagonal op, 206 cycles per op
        // on C5.  [gpu] C5 walk (116 001 ops): 9.9 ms one diagonal (32 k trips); three views, hipcc's hop code 6.7 ms; hand-written
        // hops 5.2 ms (16 k trips); five views 4.8 ms (8.2 k); both windows readable 7.5 k trips; seven views 4.6 ms (5.8 k);
        // nine views 4.7 ms (4.8 k trips, but 100 SGPRs and a longer set-up).  Timing-only builds: hops without their store
        // and exec writes -0.4 ms, no waits for the prefetched window -0.0 ms: what is left is the instruction count.
        constexpr int SR = Geo::SR;   // (for the stripe engine's power-of-two geometries every / and % below is a shift or a mask)
        constexpr uint32_t OPTAB = LOCAL ? ((uint32_t)'I' | (uint32_t)'D' << 8 | (uint32_t)'M' << 16)    // local: l 0, u 1, d 2
                                         : ((uint32_t)'D' | (uint32_t)'I' << 8 | (uint32_t)'M' << 16);   // global: u 0, l 1, d 2
        constexpr int NOCODE = 2 * WB;
#ifndef PWA_WALK_A
#define PWA_WALK_A 3
#endif
        constexpr int A = PWA_WALK_A, NV = 2 * A + 1;   // views -A .. +A
        if (lane == 0) win[NOCODE] = 0xff;
        const int dl = lane == 63 ? 0x40000000 : lane;   // lane 63 never holds a cell: every run of set mask bits ends by bit 63
        typedef unsigned long long u64;
        // The two LDS buffers form a ring over the band steps of a stripe: window w (steps 64 w .. 64 w + 63) lives in buffer
        // w & 1, so a cell at step t sits at ((t & 127) * 64 + k) * RL + r whichever window it belongs to.  The walk only moves
        // backwards: next to the window of the anchor the one before it is in flight from the moment the walk enters, and once it
        // has landed (checked a few trips later: s_waitcnt also waits for the op stores still on their way) the lanes may read
        // both -- a trip then reaches as far as its 63 lanes and five diagonals go, not just to the window's edge.
        int since = 0;
        bool pre_done = false;
        while (i > 0 && j > 0) {
            {   // make sure the window holding (i, j) is staged (wave-uniform)
                const unsigned q0 = (unsigned)(i - 1);
                const int s0 = Geo::stripe(q0), k0 = Geo::lane(Geo::row_in_stripe(q0));
                const int w0 = (int)((unsigned)(j - 1 + k0) / (unsigned)WIN);
                if (s0 != cur_s || w0 != cur_w || (NARROW && k0 < klo_of(w0 & 1))) {   // (NARROW: ... or the anchor has left the staged lanes)
                    const bool have = s0 == pre_s && w0 == pre_w && (!NARROW || k0 >= klo_of(w0 & 1));   // the window already in flight / landed
                    if (!have) {
                        issue(w0 & 1, s0, w0, k0 - DW, k0);
                        set_klo(w0 & 1, k0 - DW);
                    }
                    if (!(have && pre_done)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA is ordered for our ds_read by vmcnt
                    cur_s = s0;
                    cur_w = w0;
                    pre_s = -1;
                    pre_done = false;
                    since = 0;
                    if (w0 > 0) {
                        issue((w0 - 1) & 1, s0, w0 - 1, k0 - 2 * DW, k0);
                        set_klo((w0 - 1) & 1, k0 - 2 * DW);
                        pre_s = s0;
                        pre_w = w0 - 1;
                    }
                } else if (pre_s >= 0 && !pre_done && ++since >= 3) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    pre_done = true;
                }
            }
            // LDS addresses of this lane's cells (NOCODE when a cell is outside the matrix or the staged steps)
            const int tlo = (pre_done ? cur_w - 1 : cur_w) * WIN, jj = j - dl;
            const unsigned span = (unsigned)((cur_w + 1) * WIN - tlo);
            auto staged = [&](int t) { return (unsigned)(t - tlo) < span; };
            auto lds_at = [&](int t, int ql) { return (t & (2 * WIN - 1)) * SR + Geo::off(ql); };
            // NARROW (single-stripe pairs): rows below the anchor window's lowest staged lane count as not staged -- one range check in
            // place of the stripe compare (the window behind it holds lanes from even further down: a trip just ends a little earlier)
            const int kr = NARROW ? max(0, klo_of(cur_w & 1)) * RL : 0;
            int code[NV];
            int qlo0 = 0, t0 = 0;
            bool s0v = false;
#pragma unroll
            for (int r = 0; r <= A; ++r) {                                   // view A - r: the cell r rows above (i-d, j-d)
                const int q = i - 1 - r - dl, ql = Geo::row_in_stripe((unsigned)q);
                const bool same = NARROW ? (unsigned)(q - kr) < (unsigned)(SR - kr) && cur_s == 0
                                         : Geo::stripe((unsigned)q) == cur_s;         // same stripe (false for rows above the matrix)
                const int t = jj - 1 + Geo::lane(ql);                        // its band step
                code[A - r] = win[same && staged(t) && jj > 0 ? lds_at(t, ql) : NOCODE];
                if (r == 0) {
                    qlo0 = ql;
                    t0 = t;
                    s0v = same;
                }
            }
#pragma unroll
            for (int c = 1; c <= A; ++c)                                     // view A + c: the cell c columns left of (i-d, j-d)
                code[A + c] = win[s0v && staged(t0 - c) && jj > c ? lds_at(t0 - c, qlo0) : NOCODE];
            if (LOCAL && __builtin_amdgcn_readfirstlane(code[A]) == TB_STOP) {   // dp == 0 on the anchor, hw2.cpp:239
                stopped = true;
                break;
            }
            constexpr int CL = TbCode<LOCAL>::LEFT;
            static_assert(TbCode<LOCAL>::UP < 2 && CL < 2 && TB_DIAG == 2 && TB_STOP == 3, "the two gap codes are the values below 2");
            u64 dm[NV], gm[NV], lm[NV];   // 'd', 'u' or 'l', 'l'
            uint32_t ob[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                dm[v] = __ballot(code[v] == TB_DIAG);
                gm[v] = __ballot((unsigned)code[v] < 2u);
                lm[v] = __ballot(code[v] == CL);
                ob[v] = __builtin_amdgcn_perm(0u, OPTAB, (uint32_t)code[v]);
            }
            int di, dj, sp, sn, vtmp;
            u64 st;
            // lanes p .. p + n - 1 store their op bytes at ops[cnt ..]; exec is put back at the end of the trip (the hops between are scalar)
#define PWA_WV_A(K) "s_bfm_b64 exec, %[n], %[p]\n\t" "s_sub_u32 %[p], %[cnt], %[p]\n\t" "v_add_u32 %[tmp], %[lane], %[p]\n\t" "global_store_byte %[tmp], %[b" K "], %[ops]\n\t"
            // one hop on view K: q = end of the run of 'd' from lane p, count = run + the gap op behind it (if any)
#define PWA_WH_HOP(K)                                 \
    "s_lshr_b64 %[t], %[d" K "], %[p]\n\t"            \
    "s_not_b64 %[t], %[t]\n\t"                        \
    "s_ff1_i32_b64 %[n], %[t]\n\t"                    \
    "s_add_u32 %[q], %[p], %[n]\n\t"                  \
    "s_bitcmp1_b64 %[g" K "], %[q]\n\t"               \
    "s_addc_u32 %[n], %[n], 0\n\t"                    \
    PWA_WV_A(K)                                       \
    "s_add_u32 %[cnt], %[cnt], %[n]\n\t"
            // (below: a gap code that is not 'l' is 'u')
            // view K = A + a, a < 0, cell (i+a-p, j-p): 'l' -> view KL lane q+1, 'u' -> view KU same lane; else the trip ends |a| rows up
#define PWA_WH_NEG(K, KU, KL, RA)                                                                               \
    "LV" K "_%=:\n\t" PWA_WH_HOP(K) "s_add_u32 %[p], %[q], 1\n\t"                                               \
    "s_bitcmp1_b64 %[l" K "], %[q]\n\t"                                                                         \
    "s_cbranch_scc1 LV" KL "_%=\n\t"                                                                            \
    "s_mov_b32 %[p], %[q]\n\t"                                                                                  \
    "s_bitcmp1_b64 %[g" K "], %[q]\n\t"                                                                         \
    "s_cbranch_scc1 LV" KU "_%=\n\t"                                                                            \
    "s_mov_b32 %[dj], %[q]\n\t"                                                                                 \
    "s_add_u32 %[q], %[q], " RA "\n\t"                                                                          \
    "s_branch LE_%=\n"
            // a > 0, cell (i-p, j-a-p): 'l' -> view KL same lane, 'u' -> view KU lane q+1
#define PWA_WH_POS(K, KU, KL, CA)                                                                               \
    "LV" K "_%=:\n\t" PWA_WH_HOP(K) "s_mov_b32 %[p], %[q]\n\t"                                                  \
    "s_bitcmp1_b64 %[l" K "], %[q]\n\t"                                                                         \
    "s_cbranch_scc1 LV" KL "_%=\n\t"                                                                            \
    "s_add_u32 %[p], %[q], 1\n\t"                                                                               \
    "s_bitcmp1_b64 %[g" K "], %[q]\n\t"                                                                         \
    "s_cbranch_scc1 LV" KU "_%=\n\t"                                                                            \
    "s_add_u32 %[dj], %[q], " CA "\n\t"                                                                         \
    "s_branch LE_%=\n"
            // view A (the anchor's diagonal): 'u' / 'l' -> the neighbours, same lane
#define PWA_WH_MID(K, KU, KL)                                                                                   \
    "LV" K "_%=:\n\t" PWA_WH_HOP(K) "s_mov_b32 %[p], %[q]\n\t"                                                  \
    "s_bitcmp1_b64 %[l" K "], %[q]\n\t"                                                                         \
    "s_cbranch_scc1 LV" KL "_%=\n\t"                                                                            \
    "s_bitcmp1_b64 %[g" K "], %[q]\n\t"                                                                         \
    "s_cbranch_scc1 LV" KU "_%=\n\t"                                                                            \
    "s_mov_b32 %[dj], %[q]\n\t"                                                                                 \
    "s_branch LE_%=\n"
            // the outermost views: a 'u' (view 0) / 'l' (view 2A) leaves the staged diagonals; its op is already stored
#define PWA_WH_TOP(KL, RA)                                                                                      \
    "LV0_%=:\n\t" PWA_WH_HOP("0") "s_add_u32 %[p], %[q], 1\n\t"                                                 \
    "s_bitcmp1_b64 %[l0], %[q]\n\t"                                                                             \
    "s_cbranch_scc1 LV" KL "_%=\n\t"                                                                            \
    "s_mov_b32 %[dj], %[q]\n\t"                                                                                 \
    "s_bitcmp1_b64 %[g0], %[q]\n\t"                                                                             \
    "s_addc_u32 %[q], %[q], " RA "\n\t"                                                                         \
    "s_branch LE_%=\n"
#define PWA_WH_BOT(K, KU, CA, CA1)                                                                              \
    "LV" K "_%=:\n\t" PWA_WH_HOP(K) "s_add_u32 %[p], %[q], 1\n\t"                                               \
    "s_add_u32 %[dj], %[q], " CA1 "\n\t"                                                                        \
    "s_bitcmp1_b64 %[l" K "], %[q]\n\t"                                                                         \
    "s_cbranch_scc1 LE_%=\n\t"                                                                                  \
    "s_bitcmp1_b64 %[g" K "], %[q]\n\t"                                                                         \
    "s_cbranch_scc1 LV" KU "_%=\n\t"                                                                            \
    "s_add_u32 %[dj], %[q], " CA "\n"
#define PWA_WH_OUT [cnt] "+s"(cnt), [q] "=&s"(di), [dj] "=&s"(dj), [p] "=&s"(sp), [n] "=&s"(sn), [t] "=&s"(st), [tmp] "=&v"(vtmp)
#define PWA_WH_IN(K, V) [d##K] "s"(dm[V]), [g##K] "s"(gm[V]), [l##K] "s"(lm[V]), [b##K] "v"(ob[V])
#if PWA_WALK_A == 4
            {
                asm volatile("s_mov_b32 %[p], 0\n\t"
                             "s_branch LV4_%=\n"
                             PWA_WH_TOP("1", "4") PWA_WH_NEG("1", "0", "2", "3") PWA_WH_NEG("2", "1", "3", "2") PWA_WH_NEG("3", "2", "4", "1")
                             PWA_WH_MID("4", "3", "5")
                             PWA_WH_POS("5", "4", "6", "1") PWA_WH_POS("6", "5", "7", "2") PWA_WH_POS("7", "6", "8", "3") PWA_WH_BOT("8", "7", "4", "5")
                             "LE_%=:\n\t"
                             "s_mov_b64 exec, -1"
                             : PWA_WH_OUT
                             : PWA_WH_IN(0, 0), PWA_WH_IN(1, 1), PWA_WH_IN(2, 2), PWA_WH_IN(3, 3), PWA_WH_IN(4, 4), PWA_WH_IN(5, 5), PWA_WH_IN(6, 6),
                               PWA_WH_IN(7, 7), PWA_WH_IN(8, 8), [lane] "v"(lane), [ops] "s"(ops)
                             : "memory", "scc");
            }
#else
            {
                static_assert(A == 3, "the hop code is written out for seven or nine views");
                asm volatile("s_mov_b32 %[p], 0\n\t"
                             "s_branch LV3_%=\n"
                             PWA_WH_TOP("1", "3") PWA_WH_NEG("1", "0", "2", "2") PWA_WH_NEG("2", "1", "3", "1")
                             PWA_WH_MID("3", "2", "4")
                             PWA_WH_POS("4", "3", "5", "1") PWA_WH_POS("5", "4", "6", "2") PWA_WH_BOT("6", "5", "3", "4")
                             "LE_%=:\n\t"
                             "s_mov_b64 exec, -1"
                             : PWA_WH_OUT
                             : PWA_WH_IN(0, 0), PWA_WH_IN(1, 1), PWA_WH_IN(2, 2), PWA_WH_IN(3, 3), PWA_WH_IN(4, 4), PWA_WH_IN(5, 5), PWA_WH_IN(6, 6),
                               [lane] "v"(lane), [ops] "s"(ops)
                             : "memory", "scc");
            }
#endif
#undef PWA_WH_MID
#undef PWA_WH_TOP
#undef PWA_WH_BOT
#undef PWA_WH_OUT
#undef PWA_WH_IN
#undef PWA_WH_HOP
#undef PWA_WV_A
#undef PWA_WH_NEG
#undef PWA_WH_POS
            // the trip ended di rows and dj columns behind the anchor
            if ((di | dj) == 0) {   // cannot happen (the anchor's window is staged): never spin on the GPU
                walk_fault = true;
                break;
            }
            i -= di;
            j -= dj;
            ++best_run;   // (op-list walk: `overlap` reports the trips, for PWA_DEBUG)
        }
    }
    while (!OPS && i > 0 && j > 0) {
        {   // make sure the window holding (i, j) is staged (wave-uniform)
            const unsigned q0 = (unsigned)(i - 1);
            const int s0 = Geo::stripe(q0), k0 = Geo::lane(Geo::row_in_stripe(q0));
            const int w0 = (j - 1 + k0) / WIN;
            if (s0 != cur_s || w0 != cur_w || (NARROW && k0 < klo_of(cb))) {
                if (s0 == pre_s && w0 == pre_w && (!NARROW || k0 >= klo_of(cb ^ 1))) cb ^= 1;   // already in flight into the other buffer
                else {
                    issue(cb, s0, w0, k0 - DW, k0);
                    set_klo(cb, k0 - DW);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // LDS-DMA is ordered for our ds_read by vmcnt
                cur_s = s0;
                cur_w = w0;
                pre_s = -1;
                if (w0 > 0) {                                            // the walk only moves backwards
                    issue(cb ^ 1, s0, w0 - 1, k0 - 2 * DW, k0);
                    set_klo(cb ^ 1, k0 - 2 * DW);
                    pre_s = s0;
                    pre_w = w0 - 1;
                }
            }
        }
        const int ii = i - lane, jj = j - lane;
        auto code_at = [&](int ci, int cj) -> int {                      // code of cell (ci, cj); 0xff: outside the matrix or the staged window
            int c = 0xff;
            if (ci > 0 && cj > 0) {
                const unsigned q = (unsigned)(ci - 1);
                const int s = Geo::stripe(q), ql = Geo::row_in_stripe(q);
                const int t = cj - 1 + Geo::lane(ql);
                const int w = t / WIN;
                if (s == cur_s && w == cur_w && (!NARROW || (int)q >= klo_of(cb) * RL)) c = win[cb * WB + (t - w * WIN) * STEP_BYTES + Geo::off(ql)];
            }
            return c;
        };
        const int code = code_at(ii, jj);
        const unsigned long long dm = __ballot(code == TB_DIAG);
        const int L = (~dm == 0ull) ? 64 : __builtin_ctzll(~dm);        // leading run of diagonal moves
        if (OPS && lane < L) ops[cnt + lane] = 'M';                      // hw2.cpp:164-169 / 240-245
        if (OVL && L > 0) {
            bool eq = false;
            if (lane < L) {   // hw2.cpp:269: both symbols non-'-' and equal (a '-' inside a SEQUENCE counts as a gap column)
                const int a = pat[ii - 1];
                eq = a == (int)txt[jj - 1] && a != G.dash;
            }
            const unsigned long long em = __ballot(eq);                  // bit d: column (i-d, j-d) holds equal symbols
            const unsigned long long full = (L == 64) ? ~0ull : ((1ull << L) - 1ull);
            if (em == full) {
                run += L;
            } else {
                best_run = max(best_run, run + (int)__builtin_ctzll(~em));   // the run coming in ends inside this trip
                int k = 0;
                for (unsigned long long x = em; x; x &= x >> 1) ++k;     // longest run of ones inside the trip
                best_run = max(best_run, k);
                run = L - 1 - (63 - (int)__builtin_clzll(~em & full));   // equal columns at the far end carry on
            }
            best_run = max(best_run, run);
        }
        cnt += L;
        i -= L;
        j -= L;
        if (L < 64) {
            const int c2 = __builtin_amdgcn_readlane(code, L);
            if (c2 == TbCode<LOCAL>::UP || c2 == TbCode<LOCAL>::LEFT) {  // hw2.cpp:170-179 / 246-255
                const bool left = c2 == TbCode<LOCAL>::LEFT;
                if (OPS && lane == 0) ops[cnt] = left ? 'I' : 'D';
                ++cnt;
                i -= left ? 0 : 1;
                j -= left ? 1 : 0;
                run = 0;
                // r03: a RUN of gaps in that direction in the same trip -- the lanes look along the row (left) or the column (up) behind the
                // gap: a global alignment of a 150-row pattern against 2000 columns is mostly a few long runs of 'l', which this loop used to
                // take one op per LDS round trip ([gpu] hw2_amd -g, 262 144 pairs 150 x 2000: walks ~29 -> 17 ms of device time; what is left
                // is one 64-step window per trip, fetched at LDS-DMA latency)
                if (OVL && i > 0 && j > 0) {
                    const int rcode = left ? code_at(i, j - lane) : code_at(i - lane, j);
                    const unsigned long long gm = __ballot(rcode == c2);
                    const int R = (~gm == 0ull) ? 64 : __builtin_ctzll(~gm);
                    cnt += R;
                    i -= left ? 0 : R;
                    j -= left ? R : 0;
                }
            } else if (LOCAL && c2 == TB_STOP) {                         // dp == 0, hw2.cpp:239
                stopped = true;
                break;
            }
            // 0xff: the cell lies in another window (or outside the matrix): next trip
        }
    }
    if (!LOCAL) {
        // hw2.cpp:170-179 with j == 0 or i == 0: column 0 is all 'u', row 0 all 'l' (125-136)
        if (OPS)
            for (int o = lane; o < i; o += 64) ops[cnt + o] = 'D';
        cnt += i;
        i = 0;
        if (OPS)
            for (int o = lane; o < j; o += 64) ops[cnt + o] = 'I';
        cnt += j;
        j = 0;
    }
    (void)stopped;
    if (lane == 0) {
        res->start_i = (uint32_t)i;
        res->start_j = (uint32_t)j;
        res->n_ops = cnt;
        res->overlap = best_run;
        res->overflow = (cnt > P.ops_cap || walk_fault) ? 1u : 0u;   // cannot happen: a walk has at most n + m ops
    }
}

}  // namespace pwa
