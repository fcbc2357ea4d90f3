// postprocess.cpp -- host-side string builders over a traceback op list (no GPU work).
//
// Behavioural restatement of hw2.cpp:59-78 (CIGAR), 80-116 (MD:Z), 267-278 (overlap) and of the
// gapped-string construction inside the two alignment functions (164-184 / 240-259), written over
// flat byte buffers so that it can sit behind the C ABI (include/pwalign.h: pwa_format_alignment).
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/pwalign.h"

namespace {

inline char* put_uint(char* out, uint64_t v) {   // std::to_string of a non-negative count
    char tmp[24];
    int k = 0;
    do {
        tmp[k++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (k) *out++ = tmp[--k];
    return out;
}

}  // namespace

extern "C" {

uint64_t pwa_cigar_bound(uint64_t n_ops) { return 2 * n_ops + 24; }
uint64_t pwa_mdz_bound(uint64_t n_ops) { return 3 * n_ops + 24; }

// hw2.cpp:344 needs overlapLongestExactMatch (267-278) of EVERY pair in -g mode but the strings of the
// best pair only: this walks the op list once, without building the gapped strings.
int pwa_alignment_overlap(const uint8_t* pattern, uint64_t n, const uint8_t* text, uint64_t m, const uint8_t* ops,
                          uint64_t n_ops, const uint64_t end_cell[2], int32_t* overlap) {
    if (!end_cell || (n_ops && !ops) || !overlap) return PWA_E_INVALID;
    uint64_t i = end_cell[0], j = end_cell[1];
    if (i > n || j > m) return PWA_E_INVALID;
    int32_t best = 0, cur = 0;   // a run is the same forwards and backwards
    for (uint64_t k = 0; k < n_ops; ++k) {
        switch (ops[k]) {
            case 'M':
                if (i == 0 || j == 0) return PWA_E_INVALID;
                --i;
                --j;
                // hw2.cpp:270: both non-gap and equal; a literal '-' in a sequence never counts (it fails the != '-' test)
                if (pattern[i] == text[j] && pattern[i] != '-') {
                    if (++cur > best) best = cur;
                } else {
                    cur = 0;
                }
                break;
            case 'D':
                if (i == 0) return PWA_E_INVALID;
                --i;
                cur = 0;
                break;
            case 'I':
                if (j == 0) return PWA_E_INVALID;
                --j;
                cur = 0;
                break;
            default:
                return PWA_E_INVALID;
        }
    }
    *overlap = best;
    return PWA_OK;
}

int pwa_format_alignment(const uint8_t* pattern, uint64_t n, const uint8_t* text, uint64_t m, const uint8_t* ops,
                         uint64_t n_ops, const uint64_t end_cell[2], char* aligned_pattern, char* aligned_reference,
                         char* cigar, char* mdz, int32_t* overlap) {
    if (!end_cell || (n_ops && !ops) || !aligned_pattern || !aligned_reference) return PWA_E_INVALID;
    // ---- gapped strings: the walk emits columns end -> start; write them back to front
    uint64_t i = end_cell[0], j = end_cell[1];
    if (i > n || j > m) return PWA_E_INVALID;
    for (uint64_t k = 0; k < n_ops; ++k) {
        const uint64_t col = n_ops - 1 - k;
        switch (ops[k]) {
            case 'M':   // hw2.cpp:164-169 / 240-245
                if (i == 0 || j == 0) return PWA_E_INVALID;
                aligned_pattern[col] = (char)pattern[--i];
                aligned_reference[col] = (char)text[--j];
                break;
            case 'D':   // hw2.cpp:170-174 / 246-250
                if (i == 0) return PWA_E_INVALID;
                aligned_pattern[col] = (char)pattern[--i];
                aligned_reference[col] = '-';
                break;
            case 'I':   // hw2.cpp:175-179 / 251-255
                if (j == 0) return PWA_E_INVALID;
                aligned_pattern[col] = '-';
                aligned_reference[col] = (char)text[--j];
                break;
            default:
                return PWA_E_INVALID;
        }
    }
    aligned_pattern[n_ops] = 0;
    aligned_reference[n_ops] = 0;

    // ---- CIGAR: run-length encoding in forward order (hw2.cpp:59-78)
    if (cigar) {
        char* o = cigar;
        uint64_t k = n_ops;
        while (k > 0) {
            const uint8_t cur = ops[k - 1];
            uint64_t run = 0;
            while (k > 0 && ops[k - 1] == cur) {
                ++run;
                --k;
            }
            o = put_uint(o, run);
            *o++ = (char)cur;
        }
        *o = 0;
    }

    // ---- MD:Z (hw2.cpp:80-116)
    if (mdz) {
        char* o = mdz;
        uint64_t matches = 0, c = 0;
        while (c < n_ops) {
            const uint8_t op = ops[n_ops - 1 - c];   // forward order
            if (op == 'M') {
                if (aligned_pattern[c] == aligned_reference[c]) {
                    ++matches;
                } else {
                    o = put_uint(o, matches);
                    *o++ = aligned_reference[c];
                    matches = 0;
                }
                ++c;
            } else if (op == 'D') {
                o = put_uint(o, matches);
                *o++ = '^';
                matches = 0;
                while (c < n_ops && ops[n_ops - 1 - c] == 'D') *o++ = aligned_pattern[c++];
            } else {
                ++c;   // 'I': skipped, the match counter keeps running
            }
        }
        o = put_uint(o, matches);
        *o = 0;
    }

    // ---- longest run of identical, non-gap columns (hw2.cpp:267-278)
    if (overlap) {
        int32_t best = 0, cur = 0;
        for (uint64_t c = 0; c < n_ops; ++c) {
            const char a = aligned_pattern[c], b = aligned_reference[c];
            if (a != '-' && b != '-' && a == b) {
                if (++cur > best) best = cur;
            } else {
                cur = 0;
            }
        }
        *overlap = best;
    }
    return PWA_OK;
}

}  // extern "C"
