// pair_fill.hip.h -- full DP fill of ONE pair with the traceback band written to HBM, plus the
// traceback walk (gfx950 / MI355X).
//
// Replaces hw2.cpp:119-156 + 158-181 (NW) and hw2.cpp:193-231 + 233-257 (SW): the reference fills
// an int matrix and a char matrix row by row (5 B/cell) and walks the char matrix backwards.
//
// Mapping:
//   * the matrix is cut into horizontal STRIPES of 64*RL rows; one 64-lane wavefront owns a stripe,
//     lane k owns RL consecutive rows of it;
//   * the wave sweeps its stripe along an anti-diagonal front: at step t lane k computes column
//     c = t - k of its RL rows.  "left" and most "diag"/"up" operands are the lane's own registers;
//     the row above a lane's first row comes from lane k-1 one step earlier through a DPP
//     wave-shift (no LDS), as does the text character, which enters at lane 0 and travels down;
//   * lane 0 is fed from the stripe above through a row buffer in HBM ([column] int32), read and
//     written 64 columns at a time as one coalesced 256-B wave access;
//   * traceback codes (1 B/cell, the reference's char matrix) leave in SKEWED layout
//     tb[stripe][step][lane][RL]: every step each wave stores 64*RL contiguous bytes (256 B for
//     RL = 4); the optional int32 score band uses the same layout (1 KiB per step);
//   * SW keeps, per (lane, row-slot), the first strict maximum in column order; the final
//     reduction picks max score, then smallest i (then j is already the smallest) = the
//     reference's first row-major maximum (hw2.cpp:225-229).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pwa {

enum { TB_STOP = 0, TB_DIAG = 1, TB_UP = 2, TB_LEFT = 3 };

struct PairResult {
    int32_t score;
    uint32_t end_i, end_j;
    uint32_t start_i, start_j;
    uint32_t n_ops;
    uint32_t overflow;   // ops capacity exceeded
    uint32_t pad;
};

struct PairDesc {          // one pair of a launch
    const uint8_t* pat;   // n bytes (+ slack)
    const uint8_t* txt;   // m bytes (+ slack)
    int32_t n, m;
    uint8_t* tb;          // skewed traceback band (unused when the kernel is built without TB)
    int32_t* sband;       // optional skewed int32 score band (same indexing)
    PairResult* res;
    uint8_t* ops;         // traceback output, capacity ops_cap
    uint32_t ops_cap;
    uint32_t pad;
};

struct PairParams {
    const PairDesc* pairs;
    uint32_t n_pairs;
    uint32_t* queue;        // atomic pair counter (zeroed before every launch)
    int32_t* rowbuf;        // per workgroup: 2 x row_stride int32 (bottom row of previous / current stripe)
    uint64_t row_stride;    // >= max m + 64
    int32_t match, mismatch, gap;
};

__device__ __forceinline__ int p_addw(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int p_mulw(int a, int b) { return (int)((unsigned)a * (unsigned)b); }

// value of lane (k-1) for lane k, `fill` for lane 0
__device__ __forceinline__ int wave_shr1(int fill, int v) {
    return __builtin_amdgcn_update_dpp(fill, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

template <int RL>
__device__ __forceinline__ size_t tb_index(int i, int j, int m) {   // i, j >= 1
    const int q = i - 1;
    const int s = q / (64 * RL);
    const int k = (q % (64 * RL)) / RL;
    const int r = q % RL;
    const size_t T = (size_t)m + 63;
    return ((s * T + (size_t)(j - 1 + k)) * 64 + k) * RL + r;
}

template <int RL, bool LOCAL, bool TB, bool SBAND>
__global__ __launch_bounds__(64) void pair_fill_kernel(const PairParams G) {
  const int lane = threadIdx.x;
  int32_t* const rowbuf = G.rowbuf + (size_t)blockIdx.x * 2 * G.row_stride;
  for (;;) {
    uint32_t pid = 0;
    {
        // The electing lane id is made opaque on every trip: with a plain `lane == 0` hipcc threads
        // this branch together with a later `if (lane == 0)` of the previous trip, lane 0 then loops
        // apart from lanes 1..63 and the readfirstlane below no longer sees lane 0 (observed: hang).
        int elect = lane;
        asm volatile("" : "+v"(elect));
        if (elect == 0) pid = atomicAdd(G.queue, 1u);
    }
    pid = __builtin_amdgcn_readfirstlane(pid);
    if (pid >= G.n_pairs) break;
    const PairDesc P = G.pairs[pid];
    const int n = P.n, m = P.m;
    const int T = m + 63;
    const int n_stripes = (n + 64 * RL - 1) / (64 * RL);
    const int match = G.match, mismatch = G.mismatch, gap = G.gap;

    int bs[RL], bi[RL], bj[RL];   // SW: best per row slot
#pragma unroll
    for (int r = 0; r < RL; ++r) { bs[r] = 0; bi[r] = 0; bj[r] = 0; }

    for (int s = 0; s < n_stripes; ++s) {
        const int i_first = s * 64 * RL + lane * RL + 1;   // first row of this lane (1-based)
        int pc[RL], hl[RL];
#pragma unroll
        for (int r = 0; r < RL; ++r) {
            const int i = i_first + r;
            pc[r] = (i <= n) ? (int)P.pat[i - 1] : 256;                 // 256 never equals a text byte
            hl[r] = LOCAL ? 0 : p_mulw(i, gap);                         // dp[i][0], hw2.cpp:125-130
        }
        int diag0 = LOCAL ? 0 : p_mulw(i_first - 1, gap);               // dp[i_first-1][0]
        const int32_t* rin = rowbuf + (size_t)((s + 1) & 1) * G.row_stride;
        int32_t* rout = rowbuf + (size_t)(s & 1) * G.row_stride;
        uint8_t* tbs = TB ? P.tb + (size_t)s * T * 64 * RL : nullptr;
        int32_t* sbs = SBAND ? P.sband + (size_t)s * T * 64 * RL : nullptr;

        int bottom = 0, tch = 0;   // this lane's last-row value / text char of the previous step
        int topv = 0, tcv = 0;     // staged: top-row value and text char of column t0 + lane
        int collect = 0;           // bottom row of the stripe, column (t - 63), gathered by lane
        for (int t = 0; t < T; ++t) {
            if ((t & 63) == 0) {
                const int c = t + lane;
                if (c < m) {
                    tcv = P.txt[c];
                    topv = (s == 0) ? (LOCAL ? 0 : p_mulw(c + 1, gap)) : rin[c];   // dp[row above][c+1]
                }
            }
            const int q = t & 63;
            const int top_c = __builtin_amdgcn_readlane(topv, q);
            const int txt_c = __builtin_amdgcn_readlane(tcv, q);
            const int up_in = wave_shr1(top_c, bottom);   // dp[i_first-1][j]
            tch = wave_shr1(txt_c, tch);
            const int c = t - lane;
            uint32_t codes = 0;
            int hnew[RL];
            if (c >= 0 && c < m) {
                const int j = c + 1;
                int dg = diag0, up = up_in;
#pragma unroll
                for (int r = 0; r < RL; ++r) {
                    const int sc = (pc[r] == tch) ? match : mismatch;
                    const int tdiag = p_addw(dg, sc);
                    const int lf = hl[r];
                    const int ug = p_addw(up, gap), lg = p_addw(lf, gap);
                    int h, code;
                    if (LOCAL) {
                        h = max(0, max(tdiag, max(ug, lg)));                       // hw2.cpp:211
                        code = (h == 0) ? TB_STOP : (h == tdiag) ? TB_DIAG : (h == ug) ? TB_UP : TB_LEFT;   // 214-222
                        if (h > bs[r] && (i_first + r) <= n) {                    // hw2.cpp:225-229
                            bs[r] = h;
                            bi[r] = i_first + r;
                            bj[r] = j;
                        }
                    } else {
                        h = tdiag;                                                 // hw2.cpp:142-153
                        code = TB_DIAG;
                        if (lg > h) { h = lg; code = TB_LEFT; }
                        if (ug > h) { h = ug; code = TB_UP; }
                        if ((i_first + r) == n && j == m) P.res->score = h;      // hw2.cpp:186
                    }
                    codes |= (uint32_t)code << (8 * r);
                    dg = lf;
                    up = h;
                    hl[r] = h;
                    hnew[r] = h;
                }
                diag0 = up_in;
                bottom = up;
            } else {
#pragma unroll
                for (int r = 0; r < RL; ++r) hnew[r] = 0;
            }
            // ---- traceback band: 64*RL contiguous bytes per step
            if (!TB) {
            } else if (RL == 4) {
                reinterpret_cast<uint32_t*>(tbs)[(size_t)t * 64 + lane] = codes;
            } else {
#pragma unroll
                for (int r = 0; r < RL; ++r) tbs[((size_t)t * 64 + lane) * RL + r] = (uint8_t)(codes >> (8 * r));
            }
            if (SBAND) {
#pragma unroll
                for (int r = 0; r < RL; ++r) sbs[((size_t)t * 64 + lane) * RL + r] = hnew[r];
            }
            // ---- bottom row of the stripe: lane 63 finished column t-63 in this step
            {
                const int b63 = __builtin_amdgcn_readlane(bottom, 63);
                const int cc = t - 63;
                if (cc >= 0 && lane == (cc & 63)) collect = b63;
                if (cc >= 0 && ((cc & 63) == 63 || cc == m - 1)) {
                    const int c0 = cc & ~63;
                    if (c0 + lane <= cc) rout[c0 + lane] = collect;
                }
            }
        }
    }

    if (LOCAL) {
        // per-lane reduction over row slots, then over the wave: max score, then smallest i
        int s_best = bs[0], i_best = bi[0], j_best = bj[0];
#pragma unroll
        for (int r = 1; r < RL; ++r) {
            const bool better = bs[r] > s_best || (bs[r] == s_best && bs[r] > 0 && bi[r] < i_best);
            if (better) { s_best = bs[r]; i_best = bi[r]; j_best = bj[r]; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const int so = __shfl_xor(s_best, off), io = __shfl_xor(i_best, off), jo = __shfl_xor(j_best, off);
            const bool better = so > s_best || (so == s_best && so > 0 && io < i_best);
            if (better) { s_best = so; i_best = io; j_best = jo; }
        }
        if (lane == 0) {
            P.res->score = s_best;
            P.res->end_i = (uint32_t)i_best;
            P.res->end_j = (uint32_t)j_best;
        }
    } else if (lane == 0) {
        P.res->end_i = (uint32_t)n;
        P.res->end_j = (uint32_t)m;
    }
  }
}

// Walk of hw2.cpp:163-181 (NW) / 239-257 (SW) over the skewed band.  One lane per pair: a walk
// is a dependent chain of n+m byte reads.
template <int RL, bool LOCAL>
__global__ __launch_bounds__(64) void pair_traceback_kernel(const PairParams G) {
    const uint32_t pid = blockIdx.x * 64 + threadIdx.x;
    if (pid >= G.n_pairs) return;
    const PairDesc P = G.pairs[pid];
    const int m = P.m;
    int i = (int)P.res->end_i, j = (int)P.res->end_j;
    uint32_t cnt = 0, overflow = 0;
    for (;;) {
        int code;
        if (LOCAL) {
            if (!(i > 0 && j > 0)) break;                       // hw2.cpp:239
            code = P.tb[tb_index<RL>(i, j, m)];
            if (code == TB_STOP) break;                         // dp == 0
        } else {
            if (!(i > 0 || j > 0)) break;                       // hw2.cpp:163
            if (i > 0 && j > 0) code = P.tb[tb_index<RL>(i, j, m)];
            else code = (i > 0) ? TB_UP : TB_LEFT;              // hw2.cpp:125-136: column 0 'u', row 0 'l'
        }
        uint8_t op;
        if (code == TB_DIAG) { op = 'M'; --i; --j; }            // hw2.cpp:164-169 / 240-245
        else if (code == TB_UP) { op = 'D'; --i; }              // hw2.cpp:170-174 / 246-250
        else { op = 'I'; --j; }                                 // hw2.cpp:175-179 / 251-255
        if (cnt < P.ops_cap) P.ops[cnt] = op; else overflow = 1;
        ++cnt;
    }
    P.res->start_i = (uint32_t)i;
    P.res->start_j = (uint32_t)j;
    P.res->n_ops = cnt;
    P.res->overflow = overflow;
}

}  // namespace pwa
