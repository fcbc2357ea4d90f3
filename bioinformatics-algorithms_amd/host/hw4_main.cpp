// hw4_main.cpp -- `hw4`-compatible command line over the MI355X engine (libpwalign.so).
//
// Same surface as the reference program hw4/hw4.cpp:74-240:
//   hw4_amd -i <input.fasta> -t <tree.txt> -s <match> <mismatch> <gap>
// same argument handling (84-97), stderr texts, exit codes, FASTA rules (100-136) and output bytes (228-237).
// The all-pairs step (hw4.cpp:138-159: N(N-1)/2 Needleman-Wunsch alignments WITH traceback, then counting
// gap / mismatch columns) is ONE call of pwa_distances: the HIP kernel carries the distance of the chosen
// path through the DP (hw4's tie-break diag >= up >= left) and never stores a traceback.
// UPGMA + Newick (162-228) stay on the host (pwa_upgma_newick).
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/pwalign.h"

int main(int argc, char* argv[]) {
    if (argc < 7) {   // hw4.cpp:75-78
        std::cerr << "Usage: " << argv[0] << " -i <input.fasta> -t <tree.txt> -s <match> <mismatch> <gap>\n";
        return 1;
    }
    std::string input = "input.fasta", output = "tree.txt";
    int match = 1, mismatch = -1, gap = -1, device = 0;
    for (int i = 1; i < argc; ++i) {   // hw4.cpp:84-97: unknown options are an error here (unlike hw2)
        const std::string opt = argv[i];
        if (opt == "-i" && i + 1 < argc) input = argv[++i];
        else if (opt == "-t" && i + 1 < argc) output = argv[++i];
        else if (opt == "-s" && i + 3 < argc) {
            match = std::stoi(argv[++i]);
            mismatch = std::stoi(argv[++i]);
            gap = std::stoi(argv[++i]);
        } else if (opt == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else {
            std::cerr << "Unknown option: " << opt << '\n';
            return 1;
        }
    }
    std::ifstream in(input);
    if (!in) {   // hw4.cpp:101-104
        std::cerr << "Error opening input file: " << input << '\n';
        return 1;
    }
    std::vector<std::string> ids, seqs;
    std::string line, cur_id, cur_seq;
    while (std::getline(in, line)) {   // hw4.cpp:109-132
        if (line.empty()) continue;
        if (line.back() == '\r') line.pop_back();
        if (!line.empty() && line[0] == '>') {
            if (!cur_id.empty()) {
                ids.push_back(cur_id);
                seqs.push_back(cur_seq);
            }
            cur_id = line.substr(1);
            cur_seq.clear();
        } else {
            cur_seq += line;
        }
    }
    if (!cur_id.empty()) {
        ids.push_back(cur_id);
        seqs.push_back(cur_seq);
    }
    in.close();

    const size_t n = seqs.size();
    std::vector<double> dist(n * n, 0.0);
    if (n > 1) {
        std::string blob;
        std::vector<uint64_t> off(n + 1, 0);
        for (size_t i = 0; i < n; ++i) {
            off[i] = blob.size();
            blob += seqs[i];
        }
        off[n] = blob.size();
        std::vector<uint32_t> pa, pb;
        for (size_t i = 0; i < n; ++i)
            for (size_t j = i + 1; j < n; ++j) {   // hw4.cpp:138-141: sequence1 = i (rows), sequence2 = j (columns)
                pa.push_back((uint32_t)i);
                pb.push_back((uint32_t)j);
            }
        std::vector<int32_t> d(pa.size());
        pwa_ctx* ctx = nullptr;
        int rc = pwa_ctx_create(device, &ctx);
        if (rc != PWA_OK) {
            std::cerr << "Error: opening the MI355X device failed: " << pwa_strerror(rc) << " (no CPU fallback exists)\n";
            return 2;
        }
        rc = pwa_distances(ctx, match, mismatch, gap, reinterpret_cast<const uint8_t*>(blob.data()), off.data(), (uint32_t)n,
                           pa.data(), pb.data(), pa.size(), d.data());
        if (rc != PWA_OK) {
            std::cerr << "Error: pwa_distances failed: " << pwa_strerror(rc) << " (" << pwa_last_error(ctx) << ")\n";
            pwa_ctx_destroy(ctx);
            return 2;
        }
        pwa_ctx_destroy(ctx);
        size_t k = 0;
        for (size_t i = 0; i < n; ++i)
            for (size_t j = i + 1; j < n; ++j, ++k) dist[i * n + j] = dist[j * n + i] = (double)d[k];   // hw4.cpp:146-158
    }
    if (n == 0) {   // the reference reads clusters[0] of an empty vector here (undefined); refuse instead
        std::cerr << "Error: no sequences in " << input << '\n';
        return 1;
    }
    std::vector<const char*> names(n);
    for (size_t i = 0; i < n; ++i) names[i] = ids[i].c_str();
    uint64_t need = 0;
    pwa_upgma_newick(dist.data(), names.data(), (uint32_t)n, nullptr, 0, &need);
    std::vector<char> tree(need + 1);
    if (pwa_upgma_newick(dist.data(), names.data(), (uint32_t)n, tree.data(), tree.size(), &need) != PWA_OK) {
        std::cerr << "Error: UPGMA failed\n";
        return 2;
    }
    std::ofstream out(output);
    if (!out) {   // hw4.cpp:231-234
        std::cerr << "Error opening output file: " << output << '\n';
        return 1;
    }
    out << tree.data() << std::endl;
    out.close();
    return 0;
}
