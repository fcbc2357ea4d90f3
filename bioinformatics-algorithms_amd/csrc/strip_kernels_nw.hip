// strip_kernels_nw.hip -- instantiations of the global-alignment strip kernels (plain, gap-shifted, LANES).
#include "kernel_table.h"

namespace pwa {

#define BK(R, M, S) {R, M, S, batch_scores_kernel<R, M, S, true>, "batch_scores_kernel<R=" #R "," #M "," #S ">", nullptr, nullptr, \
                     batch_scores_kernel<R, M, S, false>}
#define BKP(R, M, S) {R, M, S, batch_scores_kernel<R, M, S, true>, "batch_scores_kernel<R=" #R "," #M "," #S ">", nullptr, nullptr, \
                      batch_scores_kernel<R, M, S, false>, batch_scores_pair_kernel<R, M, S>, \
                      batch_scores_kernel<R, M, S, true, true>, batch_scores_kernel<R, M, S, false, true>}
#define BKL(R, M, S) {R, M, S, batch_scores_kernel<R, M, S, true>, "batch_scores_kernel<R=" #R "," #M "," #S ">", nullptr, nullptr, \
                      batch_scores_kernel<R, M, S, false>, nullptr, \
                      batch_scores_kernel<R, M, S, true, true>, batch_scores_kernel<R, M, S, false, true>}
extern const BatchKernelEntry kStripKernelsNW[] = {
    BK(64, BM_NW, SC_PERM),   BK(128, BM_NW, SC_PERM),  BK(152, BM_NW, SC_PERM),
    BK(64, BM_NW, SC_CMP),    BK(128, BM_NW, SC_CMP),   BK(152, BM_NW, SC_CMP),
    BKL(64, BM_NWG, SC_PERM), BKL(128, BM_NWG, SC_PERM), BKL(152, BM_NWG, SC_PERM),
    BK(64, BM_NWG, SC_CMP),   BK(128, BM_NWG, SC_CMP),  BK(152, BM_NWG, SC_CMP),
};
extern const size_t kStripKernelsNWCount = sizeof(kStripKernelsNW) / sizeof(kStripKernelsNW[0]);

}  // namespace pwa
