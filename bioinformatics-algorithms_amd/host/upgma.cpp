// upgma.cpp -- host-side UPGMA + Newick (no GPU work): the consumer of the all-pairs distance matrix in
// hw4/hw4.cpp:162-228.  Behavioural restatement over flat buffers so that it can sit behind the C ABI
// (include/pwalign.h: pwa_upgma_newick).  Doubles and the reference's operation order throughout:
// the branch lengths are printed with std::to_string(double) ("%f"), so every rounding must agree.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/pwalign.h"

extern "C" int pwa_upgma_newick(const double* dist, const char* const* names, uint32_t n, char* out, uint64_t cap,
                                uint64_t* needed) try {
    if ((n && (!dist || !names)) || (!out && cap)) return PWA_E_INVALID;
    struct Cluster {
        int size;
        double height;
        std::string newick;
    };
    std::vector<Cluster> cl(n);
    for (uint32_t i = 0; i < n; ++i) cl[i] = {1, 0.0, names[i]};            // hw4.cpp:163-168
    size_t k = n;
    std::vector<double> d(dist, dist + (size_t)n * n);
    while (k > 1) {                                                           // hw4.cpp:170
        double best = std::numeric_limits<double>::infinity();
        size_t im = 0, jm = 0;
        for (size_t i = 0; i < k; ++i)                                        // first strict minimum (175-183)
            for (size_t j = i + 1; j < k; ++j)
                if (d[i * k + j] < best) {
                    best = d[i * k + j];
                    im = i;
                    jm = j;
                }
        Cluster mg;
        mg.size = cl[im].size + cl[jm].size;
        mg.height = best / 2.0;
        mg.newick = "(" + cl[im].newick + ":" + std::to_string(std::abs(mg.height - cl[im].height)) + "," + cl[jm].newick + ":" +
                    std::to_string(std::abs(mg.height - cl[jm].height)) + ")";   // 189-190
        const size_t nk = k - 1;
        std::vector<double> nd(nk * nk, 0.0);
        std::vector<Cluster> ncl;
        ncl.reserve(nk);
        size_t idx = 0;
        for (size_t i = 0; i < k; ++i) {
            if (i == im || i == jm) continue;
            ncl.push_back(cl[i]);
            size_t idx2 = 0;
            for (size_t j = 0; j < k; ++j) {
                if (j == im || j == jm) continue;
                nd[idx * nk + idx2++] = d[i * k + j];
            }
            const double v = (d[im * k + i] * cl[im].size + d[jm * k + i] * cl[jm].size) / mg.size;   // 219: size-weighted mean
            nd[idx * nk + (nk - 1)] = v;
            nd[(nk - 1) * nk + idx] = v;
            ++idx;
        }
        ncl.push_back(mg);
        cl.swap(ncl);
        d.swap(nd);
        k = nk;
    }
    const std::string tree = (n ? cl[0].newick : std::string()) + ":0.0;";   // 228
    if (needed) *needed = tree.size() + 1;
    if (tree.size() + 1 > cap) return PWA_E_CAPACITY;
    std::memcpy(out, tree.c_str(), tree.size() + 1);
    return PWA_OK;
} catch (...) {
    return PWA_E_NOMEM;
}
