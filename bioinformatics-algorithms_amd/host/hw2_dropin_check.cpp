// hw2_dropin_check.cpp -- calls the drop-in exactly as the reference's loop does (hw2.cpp:328-338: one call per pair, keeps
// the AlignmentResult*, deletes it afterwards 395-399) and prints the five fields, so that tests can compare them with the
// reference's.  stdin:  one case per record  "<g|l> <match> <mismatch> <gap> <n> <m>\n" + n pattern bytes + m reference bytes + "\n"
// stdout: per case  "<score> <len aligned> <len cigar> <len mdz>\n" + alignedPattern + alignedReference + cigar + mdz + "\n"
#include <cstdio>
#include <iostream>
#include <string>

#include "hw2_dropin.h"

int main() {
    char mode;
    int match, mismatch, gap;
    unsigned long n, m;
    while (std::scanf(" %c %d %d %d %lu %lu", &mode, &match, &mismatch, &gap, &n, &m) == 6) {
        if (std::getchar() != '\n') return 3;
        std::string p(n, '\0'), t(m, '\0');
        if (n && std::fread(&p[0], 1, n, stdin) != n) return 3;
        if (m && std::fread(&t[0], 1, m, stdin) != m) return 3;
        AlignmentResult* r = mode == 'g' ? globalAlignmentNeedlemanWunsch(p, t, match, mismatch, gap)
                                         : localAlignmentSmithWaterman(p, t, match, mismatch, gap);
        std::printf("%d %zu %zu %zu\n", r->score, r->alignedPattern.size(), r->cigar.size(), r->mdz.size());
        std::fwrite(r->alignedPattern.data(), 1, r->alignedPattern.size(), stdout);
        std::fwrite(r->alignedReference.data(), 1, r->alignedReference.size(), stdout);
        std::fwrite(r->cigar.data(), 1, r->cigar.size(), stdout);
        std::fwrite(r->mdz.data(), 1, r->mdz.size(), stdout);
        std::printf("\n");
        delete r;
    }
    return 0;
}
