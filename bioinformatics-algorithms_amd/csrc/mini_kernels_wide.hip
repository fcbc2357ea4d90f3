// mini_kernels_wide.hip -- the mini-stripe fills with ONE pair per wave (mini_fill.hip.h, LN = 64: single stripes of 512 / 1024 rows) and
// the walks over their bands (the stripe engine's layout with RL = 8 / 16).  Own translation unit.
#include "kernel_table.h"

namespace pwa {

template <int RL>
static pair_kernel_t mini_wide_pick(bool local, bool sband, bool gap0) {
    if (gap0) return (local || sband) ? nullptr : mini_fill_kernel<RL, false, false, true, true, 64>;
    if (local) return sband ? mini_fill_kernel<RL, true, true, false, true, 64> : mini_fill_kernel<RL, true, false, false, true, 64>;
    return sband ? mini_fill_kernel<RL, false, true, false, true, 64> : mini_fill_kernel<RL, false, false, false, true, 64>;
}
pair_kernel_t mini_wide_kernel_for(int rl, bool local, bool sband, bool gap0) {
    return rl == 6 ? mini_wide_pick<6>(local, sband, gap0) : rl == 8 ? mini_wide_pick<8>(local, sband, gap0)
           : rl == 12 ? mini_wide_pick<12>(local, sband, gap0) : rl == 16 ? mini_wide_pick<16>(local, sband, gap0) : nullptr;
}

template <int RL>
static pair_kernel_t wide_tb_pick(bool local, int walk) {
    if (local)
        return walk == WALK_OPS ? pair_traceback_kernel<RL, true, WALK_OPS, 64>
               : walk == WALK_OVERLAP ? pair_traceback_kernel<RL, true, WALK_OVERLAP, 64> : pair_traceback_kernel<RL, true, WALK_NONE, 64>;
    return walk == WALK_OPS ? pair_traceback_kernel<RL, false, WALK_OPS, 64>
           : walk == WALK_OVERLAP ? pair_traceback_kernel<RL, false, WALK_OVERLAP, 64> : pair_traceback_kernel<RL, false, WALK_NONE, 64>;
}
pair_kernel_t mini_wide_traceback_kernel_for(int rl, bool local, int walk) {
    return rl == 6 ? wide_tb_pick<6>(local, walk) : rl == 8 ? wide_tb_pick<8>(local, walk) : rl == 12 ? wide_tb_pick<12>(local, walk)
           : rl == 16 ? wide_tb_pick<16>(local, walk) : nullptr;
}

}  // namespace pwa
