// ref_shim_hw3.cpp -- TEST INFRASTRUCTURE ONLY (dev container only).
// Builds the UNMODIFIED /root/reference/Multiple_Sequence_Alignment/hw3.cpp (textual inclusion at
// compile time via -DHW3_REF_SRC=...; nothing is copied) into oracle/_ref/libhw3_ref.so and exposes
// the score pass of affine_alignment (hw3.cpp:23-102) and main (169) through a C ABI.
#define main hw3_reference_main
#include HW3_REF_SRC
#undef main

extern "C" {
int ref3_affine_score(const char* s1, size_t n, const char* s2, size_t m, int match, int mismatch, int go, int ge) {
    int score = 0;
    affine_alignment(std::string(s1, n), std::string(s2, m), match, mismatch, go, ge, &score);
    return score;
}
int ref3_main(int argc, char** argv) { return hw3_reference_main(argc, argv); }

// affine_alignment (hw3.cpp:23-135) with the strings requested; a1 / a2 are malloc'ed, NUL-terminated.
int ref3_affine_align(const char* s1, size_t n, const char* s2, size_t m, int match, int mismatch, int go, int ge, char** a1,
                      char** a2) {
    int score = 0;
    std::string x, y;
    affine_alignment(std::string(s1, n), std::string(s2, m), match, mismatch, go, ge, &score, &x, &y);
    *a1 = static_cast<char*>(std::malloc(x.size() + 1));
    *a2 = static_cast<char*>(std::malloc(y.size() + 1));
    std::copy(x.begin(), x.end(), *a1);
    std::copy(y.begin(), y.end(), *a2);
    (*a1)[x.size()] = 0;
    (*a2)[y.size()] = 0;
    return score;
}
void ref3_free(char* p) { std::free(p); }
}
