// strip_kernels_aff.hip -- instantiations of the affine (hw3) and distance (hw4) strip kernels.
#include "kernel_table.h"

namespace pwa {

extern const BatchKernelEntry kStripKernelsAff[] = {
#define AK(R, M, S, SH) {R, M, S, nullptr, "batch_affine_kernel<R=" #R "," #M "," #S ">", batch_affine_kernel<R, S, SH>}
    AK(32, BM_AFFS, SC_PERM, true),  AK(52, BM_AFFS, SC_PERM, true),  AK(32, BM_AFFS, SC_CMP, true),  AK(52, BM_AFFS, SC_CMP, true),
    AK(32, BM_AFF, SC_PERM, false),  AK(52, BM_AFF, SC_PERM, false),  AK(32, BM_AFF, SC_CMP, false),  AK(52, BM_AFF, SC_CMP, false),
#undef AK
#define DK(R, S) {R, BM_DIST, S, nullptr, "batch_nwdist_kernel<R=" #R "," #S ">", nullptr, batch_nwdist_kernel<R, S>}
    DK(32, SC_PERM), DK(64, SC_PERM), DK(32, SC_CMP), DK(64, SC_CMP),
#undef DK
#define DKP(R, S) {R, BM_DISTP, S, nullptr, "batch_nwdist_kernel<R=" #R ",PACKED>", nullptr, batch_nwdist_packed_kernel<R>}
    DKP(64, SC_PERM), DKP(128, SC_PERM), DKP(64, SC_CMP), DKP(128, SC_CMP),   // one compare-based kernel for both codings
#undef DKP
};
extern const size_t kStripKernelsAffCount = sizeof(kStripKernelsAff) / sizeof(kStripKernelsAff[0]);

}  // namespace pwa
