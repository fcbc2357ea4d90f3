// strip_kernels.hip -- the table of instantiated strip kernels the host scheduler picks from, assembled from the
// three translation units that hold the kernels (strip_kernels_{sw,nw,aff}.hip), so that they compile side by side.
#include <vector>

#include "kernel_table.h"

namespace pwa {

extern const BatchKernelEntry kStripKernelsSW[], kStripKernelsNW[], kStripKernelsAff[];
extern const size_t kStripKernelsSWCount, kStripKernelsNWCount, kStripKernelsAffCount;

static const std::vector<BatchKernelEntry>& merged_table() {
    static const std::vector<BatchKernelEntry> t = [] {
        std::vector<BatchKernelEntry> v;
        v.insert(v.end(), kStripKernelsSW, kStripKernelsSW + kStripKernelsSWCount);
        v.insert(v.end(), kStripKernelsNW, kStripKernelsNW + kStripKernelsNWCount);
        v.insert(v.end(), kStripKernelsAff, kStripKernelsAff + kStripKernelsAffCount);
        return v;
    }();
    return t;
}

const BatchKernelEntry* batch_kernel_table(size_t* count) {
    *count = merged_table().size();
    return merged_table().data();
}

const BatchKernelEntry* find_batch_kernel(int R, int mode, int score) {
    for (const auto& e : merged_table())
        if (e.R == R && e.mode == mode && e.score == score) return &e;
    return nullptr;
}

}  // namespace pwa
