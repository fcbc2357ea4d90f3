// ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (dev container only).
//
// Builds the UNMODIFIED reference translation unit
//   /root/reference/Local_Global_Alignment/hw2.cpp
// into oracle/_ref/libhw2_ref.so by textual inclusion at compile time (the path is
// given by -DHW2_REF_SRC=...; nothing of the reference is copied into this repo) and
// exposes its two alignment functions (hw2.cpp:118, 192), overlapLongestExactMatch
// (267) and main (280) through a C ABI so that tests and the golden-fixture
// generator can call the real thing.  oracle/_ref/ is git-ignored.
#define main hw2_reference_main
#include HW2_REF_SRC
#undef main

#include <cstring>

extern "C" {

struct ref_result {
    int score;
    char* aligned_pattern;
    char* aligned_reference;
    char* cigar;
    char* mdz;
};

static char* dup_str(const std::string& s) {
    char* p = static_cast<char*>(std::malloc(s.size() + 1));
    std::memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    return p;
}

static ref_result* wrap(AlignmentResult* r) {
    ref_result* o = static_cast<ref_result*>(std::malloc(sizeof(ref_result)));
    o->score = r->score;
    o->aligned_pattern = dup_str(r->alignedPattern);
    o->aligned_reference = dup_str(r->alignedReference);
    o->cigar = dup_str(r->cigar);
    o->mdz = dup_str(r->mdz);
    delete r;
    return o;
}

ref_result* ref_nw(const char* p, size_t n, const char* t, size_t m, int match, int mismatch, int gap) {
    return wrap(globalAlignmentNeedlemanWunsch(std::string(p, n), std::string(t, m), match, mismatch, gap));
}
ref_result* ref_sw(const char* p, size_t n, const char* t, size_t m, int match, int mismatch, int gap) {
    return wrap(localAlignmentSmithWaterman(std::string(p, n), std::string(t, m), match, mismatch, gap));
}
int ref_overlap(const char* ap, const char* ar) { return overlapLongestExactMatch(ap, ar); }
void ref_free(ref_result* r) {
    if (!r) return;
    std::free(r->aligned_pattern);
    std::free(r->aligned_reference);
    std::free(r->cigar);
    std::free(r->mdz);
    std::free(r);
}
int ref_main(int argc, char** argv) { return hw2_reference_main(argc, argv); }

// readFasta (hw2.cpp:25-57) -> one blob + count + 1 offsets.  The reference exit(1)s on a file it cannot
// open (28-31): callers pass existing files only.
struct ref_fasta {
    size_t count;
    char* blob;
    size_t* off;
};
ref_fasta* ref_read_fasta(const char* path) {
    const std::vector<std::string> v = readFasta(path);
    ref_fasta* f = static_cast<ref_fasta*>(std::malloc(sizeof(ref_fasta)));
    f->count = v.size();
    f->off = static_cast<size_t*>(std::malloc((v.size() + 1) * sizeof(size_t)));
    size_t tot = 0;
    for (size_t i = 0; i < v.size(); ++i) {
        f->off[i] = tot;
        tot += v[i].size();
    }
    f->off[v.size()] = tot;
    f->blob = static_cast<char*>(std::malloc(tot + 1));
    for (size_t i = 0; i < v.size(); ++i) std::memcpy(f->blob + f->off[i], v[i].data(), v[i].size());
    return f;
}
void ref_free_fasta(ref_fasta* f) {
    if (!f) return;
    std::free(f->blob);
    std::free(f->off);
    std::free(f);
}

}  // extern "C"
