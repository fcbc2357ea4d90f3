// batch_affine_tb.hip.h -- affine-gap global alignment WITH traceback for many pairs (gfx950 / MI355X).
//
// Replaces the alignments hw3.cpp builds against the center of the star
// (Multiple_Sequence_Alignment/hw3.cpp:261-283: affine_alignment(center, Si, ..., &alignedCenter, &alignedOther),
// i.e. the full hw3.cpp:23-135 with its three trace matrices and the walk 103-131).
//
// Mapping: as batch_affine.hip.h (lane = pair, register strips of R rows, 4 skewed columns per block, strip hand-off
// through HBM, atomic task queue), TRANSPOSED: the sequence the pairs share -- string1, the center -- is the wave's
// text (columns), every lane's own string2 runs down the rows.  V is symmetric under that swap and F / E change
// places, so in the reference's terms the kernel's horizontal gap state is F ('D': string1 symbol against '-') and
// its vertical one is E ('I': '-' against string2 symbol).  Values are kept UNSHIFTED, bit for bit the reference's
// V / F / E (including what it derives from its INT_MIN/2 sentinel), so every comparison below is the reference's.
//
// Per cell one code byte goes to HBM (coalesced: the four rows of a register quad are one dword per lane, a wave
// stores 256 B):
//   bits 1:0  A  = which of V / F / E is the cell's maximum in the reference's order of preference (V, then F if
//                  strictly greater, then E if strictly greater: hw3.cpp:59-68 read from the cell they index, and
//                  86-97 at the last cell): 0 = V, 1 = F, 2 = E.   traceV[i][j] == A[i-1][j-1].
//   bit  2    xF = traceF: F extended (F[i-1][j] + Ge strictly greater than V[i-1][j] + Go + Ge, 70-75)
//   bit  3    xE = traceE: E extended (77-82)
// tb layout per task: [strip][column][row quad][lane] dwords, byte b of a dword = row 4q + b of the strip.
#pragma once
#include "batch_affine.hip.h"

namespace pwa {

struct AffineTbParams {
    AffineParams a;
    uint32_t* tb;                  // code dwords of all tasks of the launch
    const uint64_t* task_tb_off;   // per task: first dword
};

// One block of C columns over the lane's R-row strip.  Vg[r] = V + Go + Ge, E[r] = horizontal state (reference F),
// D[r] = max(V, F, E) of the column left of the block on entry, of the block's last column on exit.
// Per column k: d = D of the row above in the previous column, vu / fu = (V + Go + Ge) / vertical state (reference E)
// of the row above in this column.
template <int R, int C, int SCORE>
__device__ __forceinline__ void affine_tb_block(int (&Vg)[R], int (&E)[R], int (&D)[R], const uint32_t (&pk)[R / 4],
                                                const uint32_t (&cs)[C], const int (&dtop)[C], const int (&vin)[C],
                                                const int (&fin)[C], int& topprev, int (&dbot)[C], int (&vbot)[C], int (&fbot)[C],
                                                uint32_t* tbcol /* dword of (column 0 of the block, quad 0, this lane) */,
                                                const AffineParams& P) {
    constexpr int Q = R / 4;
    int d[C], vu[C], fu[C], dl[C];
#pragma unroll
    for (int k = 0; k < C; ++k) {
        d[k] = (k == 0) ? topprev : dtop[k - 1];
        vu[k] = vin[k];
        fu[k] = fin[k];
        dl[k] = 0;
    }
    topprev = dtop[C - 1];
    const int ge = P.ge, vadd = addw(P.go, P.ge);
#pragma unroll
    for (int step = 0; step < Q + C - 1; ++step) {
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int q = step - k;
            if (q >= 0 && q < Q) {
                uint32_t s4 = 0, codes = 0;
                if (SCORE == SC_PERM) s4 = __builtin_amdgcn_perm(P.b.tab_hi, P.b.tab_lo, pk[q] ^ cs[k]);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int r = 4 * q + b;
                    int s;
                    if (SCORE == SC_PERM) s = (int)(int8_t)(s4 >> (8 * b));
                    else s = (((pk[q] >> (8 * b)) & 0xffu) == cs[k]) ? P.b.match : P.b.mismatch;
                    const int v = addw(d[k], s);                       // hw3.cpp:59-68: max(V,F,E)[i-1][j-1] + s
                    d[k] = D[r];
                    const int fext = addw(fu[k], ge);                  // vertical here = the reference's E (77-82)
                    const int xe = fext > vu[k];
                    const int f = xe ? fext : vu[k];
                    const int eext = addw(E[r], ge);                   // horizontal here = the reference's F (70-75)
                    const int xf = eext > Vg[r];
                    const int e = xf ? eext : Vg[r];
                    // the reference prefers V, then F (= e here), then E (= f here)
                    int a = 0, best = v;
                    if (e > best) { best = e; a = 1; }
                    if (f > best) { best = f; a = 2; }
                    codes |= (uint32_t)(a | (xf << 2) | (xe << 3)) << (8 * b);
                    const int vg = addw(v, vadd);
                    Vg[r] = vg;
                    E[r] = e;
                    D[r] = best;
                    vu[k] = vg;
                    fu[k] = f;
                    dl[k] = best;
                }
                tbcol[((size_t)k * Q + q) * 64] = codes;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < C; ++k) {
        dbot[k] = dl[k];
        vbot[k] = vu[k];
        fbot[k] = fu[k];
    }
}

template <int R, int SCORE>
__global__ __launch_bounds__(64, 2) void batch_affine_tb_kernel(const AffineTbParams T) {
    constexpr int Q = R / 4;
    const AffineParams& P = T.a;
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
        uint32_t* const tb = T.tb + T.task_tb_off[tid] + lane;
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
            // column 0 of the kernel = row 0 of the reference (hw3.cpp:48-53): V = F = -inf, E[0][j] = Go + Ge(j-1)
            int Vg[R], E[R], D[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                Vg[r] = addw(addw(neg, go), ge);
                E[r] = neg;
                D[r] = addw(go, mulw(row0 + r, ge));
            }
            int topprev = (s == 0) ? 0 : addw(go, mulw(row0 - 1, ge));   // V[0][0] = 0 (40), else that row's column 0

            const bool has_top = s > 0;
            const bool has_bot = s + 1 < (int)task.n_strips;
            // hand-off rows: per 4-column block three int4 per lane (D, V + Go + Ge, E-state of the strip's bottom row)
            const int32_t* hin = hand + (size_t)((s + 1) & 1) * B.hand_half;
            int32_t* hout = hand + (size_t)(s & 1) * B.hand_half;
            auto hidx = [&](int j, int which) -> size_t { return (((size_t)(j >> 2) * 192) + 64 * which + lane) * 4 + (j & 3); };
            uint32_t* tbs = tb + (size_t)s * (size_t)m * Q * 64;
            int dn = 0, vn = 0, fn = 0;
            if (has_top) {
                dn = hin[hidx(0, 0)];
                vn = hin[hidx(0, 1)];
                fn = hin[hidx(0, 2)];
            }
            // one column at a time: this pass runs for N-1 pairs next to an all-pairs score pass over N(N-1)/2
            for (int j = 0; j < m; ++j) {
                const uint32_t c = (tx[j >> 2] >> (8 * (j & 3))) & 0xffu;
                const uint32_t cs1[1] = {(SCORE == SC_PERM) ? c * 0x01010101u : c};
                // row 0 of the kernel = column 0 of the reference (42-47): V = E = -inf, F[i][0] = Go + Ge(i-1)
                const int dtop1[1] = {has_top ? dn : addw(go, mulw(j, ge))};
                const int vin1[1] = {has_top ? vn : addw(addw(neg, go), ge)};
                const int fin1[1] = {has_top ? fn : neg};
                if (has_top && j + 1 < m) {   // the next column's row above, a column ahead of its use
                    dn = hin[hidx(j + 1, 0)];
                    vn = hin[hidx(j + 1, 1)];
                    fn = hin[hidx(j + 1, 2)];
                }
                int dbot1[1], vbot1[1], fbot1[1];
                affine_tb_block<R, 1, SCORE>(Vg, E, D, pk, cs1, dtop1, vin1, fin1, topprev, dbot1, vbot1, fbot1,
                                             tbs + (size_t)j * Q * 64, P);
                if (has_bot) {
                    hout[hidx(j, 0)] = dbot1[0];
                    hout[hidx(j, 1)] = vbot1[0];
                    hout[hidx(j, 2)] = fbot1[0];
                }
            }
            {   // max(V, F, E)[n][m] (hw3.cpp:86-97) sits in this strip for the lanes whose own sequence ends here
                const int rl = n - 1 - row0;
                if (rl >= 0 && rl < R) {
                    int v = 0;
#pragma unroll
                    for (int r = 0; r < R; ++r) v = (rl == r) ? D[r] : v;
                    result = v;
                }
            }
        }
        if (outi != 0xffffffffu) B.scores[outi] = result;
    }
}

// ---------------------------------------------------------------------------------------------------
// The walk (hw3.cpp:103-131), one lane per pair, in the reference's own coordinates: i over string1 (the task's
// shared text = kernel columns), j over string2 (the lane's rows).  Emits one byte per alignment column in
// traceback order: 'M' (state V), 'D' (state F: string1 symbol against '-'), 'I' (state E).
struct AffineWalkPair {
    uint64_t tb_off;      // first dword of the pair's task
    uint64_t ops_off;     // where the pair's op list starts
    uint32_t lane;        // lane of the pair inside its task
    uint32_t n1, n2;      // lengths of string1 (columns) and string2 (rows)
    uint32_t out_index;   // pair index in the caller's list
};

template <int R>
__global__ __launch_bounds__(64) void affine_walk_kernel(const AffineWalkPair* pairs, uint32_t n_pairs, const uint32_t* tb,
                                                         uint8_t* ops, uint32_t* n_ops) {
    constexpr int Q = R / 4;
    const uint32_t p = blockIdx.x * 64 + threadIdx.x;
    if (p >= n_pairs) return;
    const AffineWalkPair W = pairs[p];
    const uint32_t* t = tb + W.tb_off + W.lane;
    const size_t m = W.n1;
    auto code = [&](uint32_t i, uint32_t j) -> uint32_t {   // interior cell (i, j >= 1)
        const uint32_t row = j - 1, s = row / R, r = row % R;
        const uint32_t w = t[(((size_t)s * m + (i - 1)) * Q + r / 4) * 64];
        return (w >> (8 * (r % 4))) & 0xffu;
    };
    auto pref = [&](uint32_t i, uint32_t j) -> int {        // A: preferred state of any cell, boundaries analytically (40-53)
        if (i == 0) return j == 0 ? 0 : 2;
        if (j == 0) return 1;
        return (int)(code(i, j) & 3u);
    };
    uint32_t i = W.n1, j = W.n2, cnt = 0;
    uint8_t* o = ops + W.ops_off;
    int state = pref(i, j);                                  // hw3.cpp:86-97
    while (i > 0 || j > 0) {
        if (state == 0) {                                    // 107-112
            o[cnt++] = 'M';
            --i;
            --j;
            state = pref(i, j);
        } else if (state == 1) {                             // 113-121
            const bool ext = (j == 0) ? (i != 1) : ((code(i, j) >> 2) & 1u);
            o[cnt++] = 'D';
            --i;
            state = ext ? 1 : 0;
        } else {                                             // 122-130
            const bool ext = (i == 0) ? (j != 1) : ((code(i, j) >> 3) & 1u);
            o[cnt++] = 'I';
            --j;
            state = ext ? 2 : 0;
        }
    }
    n_ops[W.out_index] = cnt;
}

}  // namespace pwa
