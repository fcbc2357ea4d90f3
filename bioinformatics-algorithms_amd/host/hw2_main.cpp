// hw2_main.cpp -- `hw2`-compatible command line over the MI355X engine (libpwalign.so).
//
// Same surface as the reference program (Local_Global_Alignment/hw2.cpp:280-403, README.txt:13):
//   hw2_amd -g|-l -p <patterns.fasta> -t <texts.fasta> -o <output.txt> -s <match> <mismatch> <gap>
// same argument handling (290-307), stderr texts and exit codes (281-284, 28-31, 319-322, 374-377),
// and the same bytes in the output file (379-393).  The per-pair DP (hw2.cpp:328-338) is NOT done
// here: it goes through the C ABI to the HIP kernels.  What stays on the host is what the survey
// marks as host work: FASTA reading (pwa_fasta_read: the reference's semantics, parsed in parallel into the
// engine's blob + offsets layout), CIGAR / MD:Z strings of the winner, best-pair selection, the report.
//
// Restructuring that keeps the output identical:
//   -l : the report needs only the best pair's strings and the best is chosen by score alone
//        (352-356, strict '>': first best wins) => one scores-only pass over all pairs
//        (pwa_scores) + ONE full alignment (pwa_align);
//   -g : selection needs every pair's overlap length (344-350) => pwa_overlaps over all pairs (fill + traceback
//        band + device walk that returns the overlap, no op lists) + ONE full alignment for the winner;
//   neither flag: the reference computes and then writes an EMPTY file (379-393) => no GPU work.
//
// Extra, non-colliding options: --device N (HIP device ordinal, default 0) and --devices a,b,c: the pair list is cut
// into contiguous equal blocks, one per listed GPU, each driven by its own host thread and context (pairs are
// independent, hw2.cpp:328-338; the per-pair results land in host memory, so no collective is needed in one process);
// first-best selection over the merged vector keeps the reference's loop order.
#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pwalign.h"

namespace {

struct Formatted {
    std::string aligned_pattern, aligned_reference, cigar, mdz;
    int32_t overlap = 0;
};

bool format(const std::string& p, const std::string& t, const uint8_t* ops, uint64_t n_ops, const uint64_t end[2],
            Formatted& f) {
    std::vector<char> ap(n_ops + 1), ar(n_ops + 1), cg(pwa_cigar_bound(n_ops)), md(pwa_mdz_bound(n_ops));
    if (pwa_format_alignment(reinterpret_cast<const uint8_t*>(p.data()), p.size(),
                             reinterpret_cast<const uint8_t*>(t.data()), t.size(), ops, n_ops, end, ap.data(), ar.data(),
                             cg.data(), md.data(), &f.overlap) != PWA_OK)
        return false;
    f.aligned_pattern.assign(ap.data(), n_ops);
    f.aligned_reference.assign(ar.data(), n_ops);
    f.cigar = cg.data();
    f.mdz = md.data();
    return true;
}

int engine_error(pwa_ctx* ctx, const char* what, int rc) {
    std::cerr << "Error: " << what << " failed: " << pwa_strerror(rc);
    if (ctx) std::cerr << " (" << pwa_last_error(ctx) << ")";
    std::cerr << std::endl;
    return 2;
}

}  // namespace

int main(int argc, char* argv[]) {
    if (argc < 9) {   // hw2.cpp:281-284
        std::cerr << "Usage: " << argv[0]
                  << " -g|-l -p <patterns.fasta> -t <texts.fasta> -o <output.txt> -s <match> <mismatch> <gap>" << std::endl;
        return 1;
    }
    bool global = false, local = false;
    std::string pattern_file, reference_file, output_file;
    int match = 0, mismatch = 0, gap = 0, device = 0;
    std::vector<int> devices;
    for (int i = 1; i < argc; ++i) {   // hw2.cpp:290-307; unknown tokens are ignored
        const std::string a = argv[i];
        if (a == "-g") global = true;
        else if (a == "-l") local = true;
        else if (a == "-p" && i + 1 < argc) pattern_file = argv[++i];
        else if (a == "-t" && i + 1 < argc) reference_file = argv[++i];
        else if (a == "-o" && i + 1 < argc) output_file = argv[++i];
        else if (a == "-s" && i + 3 < argc) {
            match = std::atoi(argv[++i]);
            mismatch = std::atoi(argv[++i]);
            gap = std::atoi(argv[++i]);
        } else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--devices" && i + 1 < argc) {
            const std::string list = argv[++i];
            for (size_t b = 0; b <= list.size();) {
                const size_t e = std::min(list.find(',', b), list.size());
                if (e > b) devices.push_back(std::atoi(list.substr(b, e - b).c_str()));
                b = e + 1;
            }
        }
    }
    if (devices.empty()) devices.push_back(device);
    device = devices[0];

    // r03: the device is opened (HIP initialisation, ~0.2 s of a 0.45 s run on 569 MB of FASTA) on a second thread WHILE the files are
    // parsed.  Nothing of the reference's order of errors changes: a file that cannot be opened still ends the program with its message
    // and exit code 1 before the device is ever looked at (hw2.cpp:28-31, 317-322), and a run that needs no alignment (neither -g nor
    // -l, or no pairs) never asks whether a device exists.
    pwa_ctx* early_ctx = nullptr;
    int early_rc = PWA_OK;
    std::thread opener;
    if (global || local) opener = std::thread([&] { early_rc = pwa_ctx_create(device, &early_ctx); });
    struct OpenerGuard {
        std::thread& t;
        pwa_ctx*& c;
        ~OpenerGuard() {
            if (t.joinable()) t.join();
            if (c) pwa_ctx_destroy(c);
        }
    } opener_guard{opener, early_ctx};

    // readFasta (hw2.cpp:25-57) for both files: one blob + offsets, the layout the engine takes (pwa_fasta_read).
    // The reference reads the pattern file first and exits at the first file it cannot open (28-31, 317-318).
    pwa_fasta* fa = nullptr;
    {
        const char* paths[2] = {pattern_file.c_str(), reference_file.c_str()};
        int bad = -1;
        const int rc = pwa_fasta_read(paths, 2, 0, &fa, &bad);
        if (rc == PWA_E_IO) {
            std::cerr << "Error: Cannot open file " << (bad == 1 ? reference_file : pattern_file) << std::endl;
            return 1;
        }
        if (rc != PWA_OK) return engine_error(nullptr, "reading the FASTA files", rc);
    }
    struct FastaGuard {
        pwa_fasta* f;
        ~FastaGuard() { pwa_fasta_free(f); }
    } fasta_guard{fa};
    const uint32_t* first = pwa_fasta_first_seq(fa);
    const uint64_t* off = pwa_fasta_offsets(fa);
    const uint8_t* bytes = pwa_fasta_bytes(fa);
    const size_t np = first[1] - first[0];
    if (np != (size_t)(first[2] - first[1])) {   // hw2.cpp:319-322
        std::cerr << "Error: Number of patterns and references do not match." << std::endl;
        return 1;
    }
    // sequences: patterns 0..np-1, references np..2np-1; pair i = (i, np + i)   (hw2.cpp:328-338)
    auto seq = [&](size_t k) { return std::string(reinterpret_cast<const char*>(bytes) + off[k], (size_t)(off[k + 1] - off[k])); };

    int best_index = -1;
    int32_t best_score_field = 0;
    Formatted best;

    if ((global || local) && np > 0) {
        std::vector<uint32_t> pa(np), pb(np);
        for (size_t i = 0; i < np; ++i) {
            pa[i] = (uint32_t)i;
            pb[i] = (uint32_t)(np + i);
        }

        if (opener.joinable()) opener.join();
        pwa_ctx* ctx = early_ctx;
        early_ctx = nullptr;        // owned by the code below from here on
        int rc = early_rc;
        if (rc != PWA_OK) return engine_error(nullptr, "opening the MI355X device (no CPU fallback exists)", rc);

        // the per-pair pass over contiguous blocks of the pair list, one block per listed device (block 0 on `ctx`)
        pwa_ctx* rc_ctx = nullptr;   // context whose error text belongs to a failure (kept alive until reported)
        std::vector<pwa_ctx*> extra;
        struct ExtraGuard {
            std::vector<pwa_ctx*>& v;
            ~ExtraGuard() {
                for (pwa_ctx* c : v) pwa_ctx_destroy(c);
            }
        } extra_guard{extra};
        auto sharded = [&](auto&& pass) -> int {
            const size_t D = std::min(devices.size(), np);
            if (D <= 1) return pass(ctx, (size_t)0, np);
            for (size_t d = 1; d < D; ++d) {
                pwa_ctx* c = nullptr;
                const int r = pwa_ctx_create(devices[d], &c);
                if (r != PWA_OK) return r;
                extra.push_back(c);
            }
            std::vector<int> rcs(D, PWA_OK);
            std::vector<std::thread> th;
            const size_t per = (np + D - 1) / D;
            auto run = [&](size_t d) {
                const size_t k0 = std::min(np, d * per), k1 = std::min(np, k0 + per);
                rcs[d] = k1 > k0 ? pass(d == 0 ? ctx : extra[d - 1], k0, k1) : PWA_OK;
            };
            for (size_t d = 1; d < D; ++d) th.emplace_back(run, d);
            run(0);
            for (auto& t : th) t.join();
            for (size_t d = 0; d < D; ++d)
                if (rcs[d] != PWA_OK) {
                    rc_ctx = d == 0 ? ctx : extra[d - 1];
                    return rcs[d];
                }
            return PWA_OK;
        };

        // one full alignment through the engine, rebuilt into the report's strings (both modes print only the winner)
        auto align_winner = [&](int mode, int32_t expect_score, int32_t expect_overlap) -> int {
            const std::string p = seq((size_t)best_index), t = seq(np + (size_t)best_index);
            std::vector<uint8_t> ops(p.size() + t.size() + 1);
            uint64_t n_ops = 0, end[2] = {0, 0};
            const int rc = pwa_align(ctx, mode, match, mismatch, gap, reinterpret_cast<const uint8_t*>(p.data()), p.size(),
                                     reinterpret_cast<const uint8_t*>(t.data()), t.size(), &best_score_field, ops.data(),
                                     p.size() + t.size(), &n_ops, end, nullptr);
            if (rc != PWA_OK) return engine_error(ctx, "pwa_align", rc);
            if (best_score_field != expect_score || !format(p, t, ops.data(), n_ops, end, best) ||
                (expect_overlap >= 0 && best.overlap != expect_overlap)) {
                std::cerr << "Error: inconsistent alignment for pair " << best_index << std::endl;
                return 2;
            }
            return 0;
        };

        if (global) {
            // hw2.cpp:342-350 keeps only each pair's overlap length: the device walk returns it, no op list leaves
            // the GPU (pwa_overlaps); the winner is then aligned once more for its strings
            std::vector<int32_t> scores(np), overlaps(np);
            rc = sharded([&](pwa_ctx* c, size_t k0, size_t k1) {
                return pwa_overlaps(c, PWA_MODE_NW, match, mismatch, gap, bytes, off, (uint32_t)(2 * np), pa.data() + k0,
                                    pb.data() + k0, k1 - k0, scores.data() + k0, overlaps.data() + k0);
            });
            if (rc != PWA_OK) {
                const int e = engine_error(rc_ctx ? rc_ctx : ctx, "pwa_overlaps", rc);
                pwa_ctx_destroy(ctx);
                return e;
            }
            int best_overlap = -1000000;   // hw2.cpp:326
            for (size_t i = 0; i < np; ++i)   // hw2.cpp:342-350: first strictly larger overlap wins
                if (overlaps[i] > best_overlap) {
                    best_overlap = overlaps[i];
                    best_index = (int)i;
                }
            if (const int e = align_winner(PWA_MODE_NW, scores[best_index], best_overlap)) {
                pwa_ctx_destroy(ctx);
                return e;
            }
        } else {
            std::vector<int32_t> scores(np);
            rc = sharded([&](pwa_ctx* c, size_t k0, size_t k1) {
                return pwa_scores(c, PWA_MODE_SW, match, mismatch, gap, bytes, off, (uint32_t)(2 * np), pa.data() + k0,
                                  pb.data() + k0, k1 - k0, scores.data() + k0, nullptr, nullptr);
            });
            if (rc != PWA_OK) {
                const int e = engine_error(rc_ctx ? rc_ctx : ctx, "pwa_scores", rc);
                pwa_ctx_destroy(ctx);
                return e;
            }
            int best_val = -1000000;   // hw2.cpp:326
            for (size_t i = 0; i < np; ++i)   // hw2.cpp:351-356
                if (scores[i] > best_val) {
                    best_val = scores[i];
                    best_index = (int)i;
                }
            if (const int e = align_winner(PWA_MODE_SW, scores[best_index], -1)) {
                pwa_ctx_destroy(ctx);
                return e;
            }
        }
        pwa_ctx_destroy(ctx);
    }

    std::ofstream out(output_file.c_str());   // hw2.cpp:373-377
    if (!out) {
        std::cerr << "Error: Cannot open output file " << output_file << std::endl;
        return 1;
    }
    if ((global || local) && best_index >= 0) {   // hw2.cpp:379-393; both flags: global wins
        out << (global ? "Longest overlap:" : "Highest local alignment score:") << std::endl;
        out << "pattern=" << seq((size_t)best_index) << std::endl;
        out << "reference=" << seq(np + (size_t)best_index) << std::endl;
        out << "Score =" << best_score_field << std::endl;
        out << "CIGAR =" << best.cigar << std::endl;
        out << "MD:Z=" << best.mdz << std::endl;
    }
    out.close();
    return 0;
}
