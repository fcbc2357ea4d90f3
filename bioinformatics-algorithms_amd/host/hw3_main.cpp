// hw3_main.cpp -- `hw3`-compatible command line over the MI355X engine (libpwalign.so).
//
// Same surface as the reference's center-star program (Multiple_Sequence_Alignment/hw3.cpp:169-368, README.txt:16):
//   hw3_amd -i <input.fasta> -o <output.phy> -s <M:Mm:Go:Ge>
// same argument handling and stdout messages (170-207, 139-142, 217-220, 335-338), same PHYLIP bytes (340-356).
// Both dynamic programs go through the C ABI to the HIP kernels:
//   * the all-pairs score pass hw3.cpp:232-241           -> pwa_scores_affine       (batch_affine.hip.h)
//   * the N-1 alignments against the center hw3.cpp:261-283 -> pwa_align_affine_batch (batch_affine_tb.hip.h)
// What stays on the host is string work: the FASTA reader (137-167, its quirks kept), the star sums and the choice of
// the center (230-253), the gap-pattern merge (256-328) and the PHYLIP writer.
//
// One extra, non-colliding option: --device N (HIP device ordinal, default 0).
#include <cctype>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pwalign.h"

namespace {

// hw3.cpp:137-167.  Headers are kept verbatim after '>' (a trailing '\r' included); white space is dropped anywhere in
// a sequence line; lines are skipped only when EMPTY; a record is stored only once a NON-EMPTY header has been seen.
bool read_fasta(const std::string& path, std::vector<std::pair<std::string, std::string>>& out) {
    std::ifstream in(path.c_str(), std::ios::binary);
    if (!in) return false;
    std::string line, header, sequence;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        if (line[0] == '>') {
            if (!header.empty()) {
                out.push_back({header, sequence});
                sequence.clear();
            }
            header = line.substr(1);
        } else {
            for (const char c : line)
                if (!std::isspace((unsigned char)c)) sequence.push_back(c);
        }
    }
    if (!header.empty()) out.push_back({header, sequence});
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
    using std::cout;
    using std::endl;
    using std::string;
    int device = 0;
    {   // --device N is taken out of argv before the reference's own parsing sees it
        std::vector<char*> keep;
        for (int i = 0; i < argc; ++i) {
            if (i > 0 && string(argv[i]) == "--device" && i + 1 < argc) {
                device = std::atoi(argv[++i]);
                continue;
            }
            keep.push_back(argv[i]);
        }
        argc = (int)keep.size();
        for (int i = 0; i < argc; ++i) argv[i] = keep[(size_t)i];
    }
    if (argc < 7) {   // hw3.cpp:170-173
        cout << "Usage: " << argv[0] << " -i input.fasta -o output.phy -s matchScore:mismatchScore:gapOpeningScore:gapExtensionScore"
             << endl;
        return 0;
    }
    string input_file, output_file, scores_str;
    for (int i = 1; i < argc; ++i) {   // hw3.cpp:176-188
        const string arg = argv[i];
        if (arg == "-i" && i + 1 < argc) input_file = argv[++i];
        else if (arg == "-o" && i + 1 < argc) output_file = argv[++i];
        else if (arg == "-s" && i + 1 < argc) scores_str = argv[++i];
        else {
            cout << "Unknown argument: " << arg << endl;
            return 0;
        }
    }
    int match = 0, mismatch = 0, gap_open = 0, gap_extend = 0;
    {   // hw3.cpp:191-207
        std::vector<int> values;
        std::stringstream ss(scores_str);
        string token;
        try {
            while (std::getline(ss, token, ':')) values.push_back(std::stoi(token));
        } catch (const std::exception&) {
            std::cerr << "Error: a score is not a number" << endl;   // the reference dies of the uncaught exception
            return 134;
        }
        if (values.size() != 4) {
            cout << "Error: Score must have four values separated by ':'" << endl;
            return 0;
        }
        match = values[0];
        mismatch = values[1];
        gap_open = values[2];
        gap_extend = values[3];
    }

    std::vector<std::pair<string, string>> fasta;
    if (!read_fasta(input_file, fasta)) {   // hw3.cpp:139-142 (exit(1) inside readFASTA)
        cout << "Error: Could not open file " << input_file << endl;
        return 1;
    }
    const size_t N = fasta.size();
    if (N == 0) {   // hw3.cpp:217-220
        cout << "No sequences found in " << input_file << endl;
        return 0;
    }
    if (N == 1) {   // hw3.cpp:223-230
        std::ofstream out(output_file.c_str());
        out << "1 " << fasta[0].second.size() << "\n";
        out << fasta[0].first << " " << fasta[0].second << "\n";
        out.close();
        return 0;
    }

    // sequences as one blob + offsets (the engine's input layout)
    string blob;
    std::vector<uint64_t> off(N + 1, 0);
    for (size_t i = 0; i < N; ++i) {
        off[i] = blob.size();
        blob += fasta[i].second;
    }
    off[N] = blob.size();
    const uint8_t* bytes = reinterpret_cast<const uint8_t*>(blob.data());

    pwa_ctx* ctx = nullptr;
    int rc = pwa_ctx_create(device, &ctx);
    if (rc != PWA_OK) return engine_error(nullptr, "opening the MI355X device (no CPU fallback exists)", rc);

    // ---- star scores over all pairs i < j (hw3.cpp:232-243) and the center (246-253: first strict maximum)
    size_t center = 0;
    {
        std::vector<uint32_t> pa, pb;
        pa.reserve(N * (N - 1) / 2);
        pb.reserve(N * (N - 1) / 2);
        for (size_t i = 0; i < N; ++i)
            for (size_t j = i + 1; j < N; ++j) {
                pa.push_back((uint32_t)i);
                pb.push_back((uint32_t)j);
            }
        std::vector<int32_t> pair_score(pa.size());
        rc = pwa_scores_affine(ctx, match, mismatch, gap_open, gap_extend, bytes, off.data(), (uint32_t)N, pa.data(), pb.data(),
                               pa.size(), pair_score.data());
        if (rc != PWA_OK) {
            const int e = engine_error(ctx, "pwa_scores_affine", rc);
            pwa_ctx_destroy(ctx);
            return e;
        }
        std::vector<int> sum(N, 0);   // int, as the reference
        for (size_t k = 0; k < pa.size(); ++k) {
            sum[pa[k]] = (int)((unsigned)sum[pa[k]] + (unsigned)pair_score[k]);
            sum[pb[k]] = (int)((unsigned)sum[pb[k]] + (unsigned)pair_score[k]);
        }
        for (size_t i = 1; i < N; ++i)
            if (sum[i] > sum[center]) center = i;
    }

    // ---- the N-1 alignments against the center (hw3.cpp:261-283)
    const string& cseq = fasta[center].second;
    std::vector<string> aligned_center(N), aligned_other(N);
    {
        std::vector<uint32_t> pa, pb, who;
        std::vector<uint64_t> ops_off;
        uint64_t tot = 0;
        for (size_t i = 0; i < N; ++i) {
            if (i == center) continue;
            pa.push_back((uint32_t)center);
            pb.push_back((uint32_t)i);
            who.push_back((uint32_t)i);
            ops_off.push_back(tot);
            tot += cseq.size() + fasta[i].second.size();
        }
        std::vector<uint8_t> ops(tot + 1);
        std::vector<int32_t> score(pa.size());
        std::vector<uint64_t> n_ops(pa.size());
        rc = pwa_align_affine_batch(ctx, match, mismatch, gap_open, gap_extend, bytes, off.data(), (uint32_t)N, pa.data(),
                                    pb.data(), pa.size(), score.data(), ops.data(), ops_off.data(), n_ops.data());
        if (rc != PWA_OK) {
            const int e = engine_error(ctx, "pwa_align_affine_batch", rc);
            pwa_ctx_destroy(ctx);
            return e;
        }
        for (size_t k = 0; k < pa.size(); ++k) {   // op lists (traceback order) -> the two gapped strings (133-134)
            const string& other = fasta[who[k]].second;
            string& ac = aligned_center[who[k]];
            string& ao = aligned_other[who[k]];
            size_t p1 = 0, p2 = 0;
            for (uint64_t c = n_ops[k]; c-- > 0;) {
                const uint8_t op = ops[ops_off[k] + c];
                if ((op != 'I' && p1 >= cseq.size()) || (op != 'D' && p2 >= other.size())) {
                    std::cerr << "Error: inconsistent traceback for sequence " << who[k] << std::endl;
                    pwa_ctx_destroy(ctx);
                    return 2;
                }
                ac.push_back(op == 'I' ? '-' : cseq[p1++]);
                ao.push_back(op == 'D' ? '-' : other[p2++]);
            }
            if (p1 != cseq.size() || p2 != other.size()) {
                std::cerr << "Error: inconsistent traceback for sequence " << who[k] << std::endl;
                pwa_ctx_destroy(ctx);
                return 2;
            }
        }
    }
    pwa_ctx_destroy(ctx);

    // ---- gap patterns and their merge (hw3.cpp:256-291)
    const size_t L = cseq.size();
    std::vector<std::vector<int>> gap(N, std::vector<int>(L + 1, 0));
    for (size_t i = 0; i < N; ++i) {
        if (i == center) continue;
        size_t pos = 0;
        for (const char c : aligned_center[i]) {
            if (c == '-') ++gap[i][pos];
            else ++pos;
        }
    }
    std::vector<int> merged(L + 1, 0);
    for (size_t i = 0; i < N; ++i)
        for (size_t k = 0; k <= L; ++k) merged[k] = std::max(merged[k], gap[i][k]);

    // ---- final rows (hw3.cpp:293-328)
    std::vector<string> fin(N);
    for (size_t k = 0; k <= L; ++k) {
        fin[center].append((size_t)(merged[k] - gap[center][k]), '-');
        if (k < L) fin[center].push_back(cseq[k]);
    }
    for (size_t i = 0; i < N; ++i) {
        if (i == center) continue;
        size_t fi = 0, oi = 0;
        while (fi < fin[center].size() && oi < aligned_center[i].size()) {
            if (fin[center][fi] == aligned_center[i][oi]) {
                fin[i].push_back(aligned_other[i][oi]);
                ++fi;
                ++oi;
            } else if (fin[center][fi] == '-') {
                fin[i].push_back('-');
                ++fi;
            } else {
                fin[i].push_back(aligned_other[i][oi]);
                ++oi;
            }
        }
        while (fi < fin[center].size()) {
            fin[i].push_back('-');
            ++fi;
        }
    }
    std::swap(fin[0], fin[center]);   // hw3.cpp:331-332: the center goes first
    std::swap(fasta[0], fasta[center]);

    std::ofstream out(output_file.c_str());   // hw3.cpp:334-338
    if (!out) {
        cout << "Error: Could not open output file " << output_file << endl;
        return 0;
    }
    out << N << " " << fin[0].size() << "\n";   // hw3.cpp:340-356
    for (size_t i = 0; i < N; ++i) {
        string id = fasta[i].first;
        if (id.size() > 10) id = id.substr(0, 10);
        else if (id.size() < 10) id.append(10 - id.size(), ' ');
        out << id;
        out << fin[i][0];   // (an empty row prints its terminator, as the reference's operator[] does)
        for (size_t j = 1; j < fin[i].size(); ++j) {
            if (j % 10 == 0) out << " ";
            out << fin[i][j];
        }
        out << "\n";
    }
    out.close();
    return 0;
}
