// strip_kernels_sw.hip -- instantiations of the local-alignment strip kernels (plain, saturating, paired, LANES).
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
extern const BatchKernelEntry kStripKernelsSW[] = {
    BK(76, BM_SW, SC_PERM),   BK(104, BM_SW, SC_PERM),
    BKL(40, BM_SWS, SC_PERM), BKP(52, BM_SWS, SC_PERM), BKP(76, BM_SWS, SC_PERM), BKL(96, BM_SWS, SC_PERM),   // R=96 paired spills in the column loop
    BKL(40, BM_SWS, SC_CMP),  BKP(52, BM_SWS, SC_CMP),  BKP(76, BM_SWS, SC_CMP),  BKP(96, BM_SWS, SC_CMP),
    BK(64, BM_SW, SC_PERM),   BK(128, BM_SW, SC_PERM),  BK(152, BM_SW, SC_PERM),
    BK(64, BM_SW, SC_CMP),    BK(128, BM_SW, SC_CMP),   BK(152, BM_SW, SC_CMP),
};
extern const size_t kStripKernelsSWCount = sizeof(kStripKernelsSW) / sizeof(kStripKernelsSW[0]);

}  // namespace pwa
