// hw2_dropin.cpp -- INTEGRATION.md Option B as a compiled unit: the two functions the reference's pair loop calls
// (hw2.cpp:331-335), with the reference's signatures and ownership rules, on the MI355X engine.
// One pwa_ctx per calling thread (contexts are not thread-safe), created on first use, destroyed at thread exit.
#include "hw2_dropin.h"

#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "../../include/pwalign.h"

namespace {

int g_device = 0;

struct ThreadCtx {
    pwa_ctx* ctx = nullptr;
    ~ThreadCtx() { pwa_ctx_destroy(ctx); }
};

pwa_ctx* context() {
    thread_local ThreadCtx tc;
    if (!tc.ctx) {
        const int rc = pwa_ctx_create(g_device, &tc.ctx);
        if (rc != PWA_OK) {
            std::cerr << "Error: opening the MI355X device failed: " << pwa_strerror(rc) << " (no CPU fallback exists)" << std::endl;
            std::exit(2);
        }
    }
    return tc.ctx;
}

AlignmentResult* align_on_device(int mode, const std::string& p, const std::string& t, int match, int mismatch, int gap) {
    pwa_ctx* ctx = context();
    std::vector<uint8_t> ops(p.size() + t.size() + 1);
    int32_t score = 0;
    uint64_t n_ops = 0, end[2] = {0, 0};
    int rc = pwa_align(ctx, mode, match, mismatch, gap, reinterpret_cast<const uint8_t*>(p.data()), p.size(),
                       reinterpret_cast<const uint8_t*>(t.data()), t.size(), &score, ops.data(), p.size() + t.size(), &n_ops, end, nullptr);
    if (rc != PWA_OK) {
        std::cerr << "Error: pwa_align failed: " << pwa_strerror(rc) << " (" << pwa_last_error(ctx) << ")" << std::endl;
        std::exit(2);
    }
    std::vector<char> ap(n_ops + 1), ar(n_ops + 1), cg(pwa_cigar_bound(n_ops)), md(pwa_mdz_bound(n_ops));
    rc = pwa_format_alignment(reinterpret_cast<const uint8_t*>(p.data()), p.size(), reinterpret_cast<const uint8_t*>(t.data()), t.size(),
                              ops.data(), n_ops, end, ap.data(), ar.data(), cg.data(), md.data(), nullptr);
    if (rc != PWA_OK) {
        std::cerr << "Error: pwa_format_alignment failed: " << pwa_strerror(rc) << std::endl;
        std::exit(2);
    }
    AlignmentResult* r = new AlignmentResult();   // hw2.cpp:122 / 199; deleted by the caller (395-399)
    r->score = score;
    r->alignedPattern.assign(ap.data(), n_ops);   // the sequences are raw bytes: a NUL inside one must survive
    r->alignedReference.assign(ar.data(), n_ops);
    r->cigar = cg.data();
    r->mdz = md.data();
    return r;
}

}  // namespace

void hw2DropinSetDevice(int device) { g_device = device; }

AlignmentResult* globalAlignmentNeedlemanWunsch(const std::string& patterns, const std::string& references, int matchScore,
                                                int mismatchScore, int gapPenalty) {
    return align_on_device(PWA_MODE_NW, patterns, references, matchScore, mismatchScore, gapPenalty);
}

AlignmentResult* localAlignmentSmithWaterman(const std::string& patterns, const std::string& references, int matchScore,
                                             int mismatchScore, int gapPenalty) {
    return align_on_device(PWA_MODE_SW, patterns, references, matchScore, mismatchScore, gapPenalty);
}
