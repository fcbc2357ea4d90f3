// mini_walk_kernels.hip -- the traceback walks (pair_fill.hip.h: pair_traceback_kernel) over the mini-stripe engine's band
// geometry (BandGeo<16, RL>).  Own translation unit.
#include "kernel_table.h"

namespace pwa {

template <int RL>
static pair_kernel_t mini_tb_pick(bool local, int walk) {
    if (local)
        return walk == WALK_OPS ? pair_traceback_kernel<RL, true, WALK_OPS, 16>
               : walk == WALK_OVERLAP ? pair_traceback_kernel<RL, true, WALK_OVERLAP, 16> : pair_traceback_kernel<RL, true, WALK_NONE, 16>;
    return walk == WALK_OPS ? pair_traceback_kernel<RL, false, WALK_OPS, 16>
           : walk == WALK_OVERLAP ? pair_traceback_kernel<RL, false, WALK_OVERLAP, 16> : pair_traceback_kernel<RL, false, WALK_NONE, 16>;
}
pair_kernel_t mini_wide_traceback_kernel_for(int rl, bool local, int walk);   // mini_kernels_wide.hip
pair_kernel_t mini_traceback_kernel_for(int rl, bool local, int walk, int ln) {
    if (ln == 64) return mini_wide_traceback_kernel_for(rl, local, walk);
    switch (rl) {
        case 4: return mini_tb_pick<4>(local, walk);
        case 6: return mini_tb_pick<6>(local, walk);
        case 8: return mini_tb_pick<8>(local, walk);
        case 10: return mini_tb_pick<10>(local, walk);
        case 12: return mini_tb_pick<12>(local, walk);
        case 16: return mini_tb_pick<16>(local, walk);
        default: return nullptr;
    }
}

}  // namespace pwa
