// kernel_table.h -- the instantiated strip kernels (strip_kernels.hip) and pair-engine kernels (pair_kernels.hip) as
// seen by the host scheduler (pwalign.hip).  Three translation units so that the device code compiles in parallel.
#pragma once
#include "batch_scores.hip.h"
#include "batch_affine.hip.h"
#include "batch_nwdist.hip.h"
#include "pair_fill.hip.h"
#include "mini_fill.hip.h"

namespace pwa {

typedef void (*batch_kernel_t)(const BatchParams);
typedef void (*affine_kernel_t)(const AffineParams);
typedef void (*nwdist_kernel_t)(const NwDistParams);
typedef void (*pair_kernel_t)(const PairParams);
enum { BM_AFF = 3, BM_AFFS = 4, BM_DIST = 5, BM_DISTP = 7 };   // affine (hw3) plain / shifted, hw4 NW + distance, its packed-key form; 0..2, 6: batch_scores.hip.h
struct BatchKernelEntry {
    int R, mode, score;
    batch_kernel_t fn;       // multi-strip form (strip hand-off rows through HBM)
    const char* name;
    affine_kernel_t afn = nullptr;
    nwdist_kernel_t dfn = nullptr;
    batch_kernel_t fn_single = nullptr;   // every task a single strip: no hand-off accesses at all
    batch_kernel_t fn_pair = nullptr;     // every task at most two strips: two waves per task, hand-off through an LDS ring
    batch_kernel_t fn_lanes = nullptr;         // every lane its own text (index-paired lists), multi-strip
    batch_kernel_t fn_lanes_single = nullptr;  // ... every task a single strip
};
// strip_kernels.hip
const BatchKernelEntry* batch_kernel_table(size_t* count);
const BatchKernelEntry* find_batch_kernel(int R, int mode, int score);
// pair_kernels.hip
// perm: coded sequences, table scoring (keyed tb only); keyed = false: the plain int32 traceback form (RL = 4 only)
// gap0: global fill in gap-shifted coordinates (the host passes gap 0 and scores s - 2 gap): perm && keyed && !sband only
pair_kernel_t pair_fill_kernel_for(int rl, int w, bool local, bool tb, bool sband, bool perm, bool keyed = true, bool gap0 = false, bool band = true);   // band = false: keyed, table scoring, no band at all
pair_kernel_t pair_traceback_kernel_for(int rl, bool local, int walk);   // walk: WALK_NONE / WALK_OPS / WALK_OVERLAP
// mini_kernels*.hip -- the mini-stripe engine (16 lanes per pair, 4 pairs per wave; keyed cells, table scoring): fills for
// rl in kMiniRL; gap0 only global without score band; the walks over its band geometry (BandGeo<16, rl>)
constexpr int kMiniRL[] = {4, 6, 8, 10, 12, 16};
// ln = 16: four pairs per wave (rl in kMiniRL); ln = 64: one pair per wave, rl = 8 | 16 (single stripes of 512 / 1024 rows; band only)
pair_kernel_t mini_fill_kernel_for(int rl, bool local, bool sband, bool gap0, bool band = true, int ln = 16);   // band = false: scores (+ end cells) only
pair_kernel_t mini_traceback_kernel_for(int rl, bool local, int walk, int ln = 16);

}  // namespace pwa
