// mini_kernels.hip -- every fill instantiation of the mini-stripe engine (mini_fill.hip.h).  Own translation unit.
#include "kernel_table.h"

namespace pwa {

template <int RL>
static pair_kernel_t mini_fill_pick(bool local, bool sband, bool gap0) {
    if (gap0) return (local || sband) ? nullptr : mini_fill_kernel<RL, false, false, true>;
    if (local) return sband ? mini_fill_kernel<RL, true, true, false> : mini_fill_kernel<RL, true, false, false>;
    return sband ? mini_fill_kernel<RL, false, true, false> : mini_fill_kernel<RL, false, false, false>;
}
pair_kernel_t mini_scores_kernel_for(int rl, bool local, bool gap0);   // mini_kernels_noband.hip
pair_kernel_t mini_wide_kernel_for(int rl, bool local, bool sband, bool gap0);   // mini_kernels_wide.hip
pair_kernel_t mini_fill_kernel_for(int rl, bool local, bool sband, bool gap0, bool band, int ln) {
    if (ln == 64) return band ? mini_wide_kernel_for(rl, local, sband, gap0) : nullptr;
    if (!band) return sband ? nullptr : mini_scores_kernel_for(rl, local, gap0);
    switch (rl) {
        case 4: return mini_fill_pick<4>(local, sband, gap0);
        case 6: return mini_fill_pick<6>(local, sband, gap0);
        case 8: return mini_fill_pick<8>(local, sband, gap0);
        case 10: return mini_fill_pick<10>(local, sband, gap0);
        case 12: return mini_fill_pick<12>(local, sband, gap0);
        case 16: return mini_fill_pick<16>(local, sband, gap0);
        default: return nullptr;
    }
}

}  // namespace pwa
