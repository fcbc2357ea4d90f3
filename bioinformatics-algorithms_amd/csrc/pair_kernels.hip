// pair_kernels.hip -- every instantiation of the wavefront (pair) engine: fill kernels and traceback walks.
// Own translation unit: compiles next to pwalign.hip and strip_kernels.hip.
#include "kernel_table.h"

namespace pwa {

template <int RL, int W>
static pair_kernel_t pair_fill_pick(bool local, bool tb, bool sband, bool perm) {
    if (tb && perm) {   // coded sequences: byte-table scoring
        if (local) return sband ? pair_fill_kernel<RL, W, true, true, true, true> : pair_fill_kernel<RL, W, true, true, false, true>;
        return sband ? pair_fill_kernel<RL, W, false, true, true, true> : pair_fill_kernel<RL, W, false, true, false, true>;
    }
    if (local) {
        if (tb) return sband ? pair_fill_kernel<RL, W, true, true, true> : pair_fill_kernel<RL, W, true, true, false>;
        return pair_fill_kernel<RL, W, true, false, false>;
    }
    if (tb) return sband ? pair_fill_kernel<RL, W, false, true, true> : pair_fill_kernel<RL, W, false, true, false>;
    return pair_fill_kernel<RL, W, false, false, false>;
}
// traceback fill in plain int32 (compare-and-select chains, no packed keys): any scores the reference's `int` holds.
// Rare (scores x lengths beyond 2^28), so only the RL = 4 geometry is instantiated.
template <int W>
static pair_kernel_t pair_fill_pick_plain_tb(bool local, bool sband) {
    if (local) return sband ? pair_fill_kernel<4, W, true, true, true, false, false> : pair_fill_kernel<4, W, true, true, false, false, false>;
    return sband ? pair_fill_kernel<4, W, false, true, true, false, false> : pair_fill_kernel<4, W, false, true, false, false, false>;
}
// the keyed chunk without a band: scores (and end cells) of long pairs over a coded arena (pwalign.hip, batch_create_impl)
template <int RL, int W>
static pair_kernel_t pair_fill_pick_noband(bool local, bool gap0) {
    if (local) return pair_fill_kernel<RL, W, true, true, false, true, true, false, false>;
    return gap0 ? pair_fill_kernel<RL, W, false, true, false, true, true, true, false> : pair_fill_kernel<RL, W, false, true, false, true, true, false, false>;
}
pair_kernel_t pair_fill_kernel_for(int rl, int w, bool local, bool tb, bool sband, bool perm, bool keyed, bool gap0, bool band) {
    if (!band) {
        if (!tb || sband || !perm || !keyed) return nullptr;
        if (rl == 2) return w == 1 ? pair_fill_pick_noband<2, 1>(local, gap0) : pair_fill_pick_noband<2, 4>(local, gap0);
        return w == 1 ? pair_fill_pick_noband<4, 1>(local, gap0) : pair_fill_pick_noband<4, 4>(local, gap0);
    }
    if (gap0) {   // gap-shifted global fills: table scoring, keyed, no score band
        if (local || !tb || sband || !perm || !keyed) return nullptr;
        if (rl == 2) return w == 1 ? pair_fill_kernel<2, 1, false, true, false, true, true, true> : pair_fill_kernel<2, 4, false, true, false, true, true, true>;
        return w == 1 ? pair_fill_kernel<4, 1, false, true, false, true, true, true> : pair_fill_kernel<4, 4, false, true, false, true, true, true>;
    }
    if (tb && !keyed) return rl != 4 ? nullptr : (w == 1 ? pair_fill_pick_plain_tb<1>(local, sband) : pair_fill_pick_plain_tb<4>(local, sband));
    if (rl == 2) return w == 1 ? pair_fill_pick<2, 1>(local, tb, sband, perm) : pair_fill_pick<2, 4>(local, tb, sband, perm);
    return w == 1 ? pair_fill_pick<4, 1>(local, tb, sband, perm) : pair_fill_pick<4, 4>(local, tb, sband, perm);
}
template <int RL>
static pair_kernel_t pair_tb_pick(bool local, int walk) {
    if (local)
        return walk == WALK_OPS ? pair_traceback_kernel<RL, true, WALK_OPS>
               : walk == WALK_OVERLAP ? pair_traceback_kernel<RL, true, WALK_OVERLAP> : pair_traceback_kernel<RL, true, WALK_NONE>;
    return walk == WALK_OPS ? pair_traceback_kernel<RL, false, WALK_OPS>
           : walk == WALK_OVERLAP ? pair_traceback_kernel<RL, false, WALK_OVERLAP> : pair_traceback_kernel<RL, false, WALK_NONE>;
}
pair_kernel_t pair_traceback_kernel_for(int rl, bool local, int walk) {
    return rl == 2 ? pair_tb_pick<2>(local, walk) : pair_tb_pick<4>(local, walk);
}

}  // namespace pwa
