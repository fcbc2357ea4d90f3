// ref_shim_hw4.cpp -- TEST INFRASTRUCTURE ONLY (dev container only).
// Builds the UNMODIFIED /root/reference/hw4/hw4.cpp (textual inclusion at compile time; nothing is
// copied) and exposes needleman_wunsch + the distance rule of main (hw4.cpp:146-152) through a C ABI.
#define main hw4_reference_main
#include HW4_REF_SRC
#undef main
extern "C" {
int ref4_nw_distance(const char* s1, size_t n, const char* s2, size_t m, int match, int mismatch, int gap) {
    auto r = needleman_wunsch(std::string(s1, n), std::string(s2, m), match, mismatch, gap);
    int d = 0;
    for (size_t k = 0; k < r.first->size(); ++k) {
        if ((*r.first)[k] == '-' || (*r.second)[k] == '-') ++d;
        else if ((*r.first)[k] != (*r.second)[k]) ++d;
    }
    delete r.first;
    delete r.second;
    return d;
}
int ref4_main(int argc, char** argv) { return hw4_reference_main(argc, argv); }
}
