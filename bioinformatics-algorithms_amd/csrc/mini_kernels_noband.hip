// mini_kernels_noband.hip -- the mini-stripe fills WITHOUT a band (mini_fill.hip.h, BAND = false): scores and end cells of
// short-pattern pairs inside a scores pass.  Own translation unit.
#include "kernel_table.h"

namespace pwa {

template <int RL>
static pair_kernel_t mini_scores_pick(bool local, bool gap0) {
    if (local) return gap0 ? nullptr : mini_fill_kernel<RL, true, false, false, false>;
    return gap0 ? mini_fill_kernel<RL, false, false, true, false> : mini_fill_kernel<RL, false, false, false, false>;
}
pair_kernel_t mini_scores_kernel_for(int rl, bool local, bool gap0) {
    switch (rl) {
        case 4: return mini_scores_pick<4>(local, gap0);
        case 6: return mini_scores_pick<6>(local, gap0);
        case 8: return mini_scores_pick<8>(local, gap0);
        case 10: return mini_scores_pick<10>(local, gap0);
        case 12: return mini_scores_pick<12>(local, gap0);
        case 16: return mini_scores_pick<16>(local, gap0);
        default: return nullptr;
    }
}

}  // namespace pwa
