// hw2_dropin.h -- the reference's own per-pair interface (Local_Global_Alignment/hw2.cpp:17-23, 118, 192), implemented over
// libpwalign.so.  A maintainer of hw2.cpp deletes the struct and the two function bodies (hw2.cpp:17-23, 118-190, 192-265),
// includes this header and links libhw2_dropin.so: main (280-403) and everything it prints stay as they are.
#pragma once
#include <string>

struct AlignmentResult {   // hw2.cpp:17-23: same fields, same order, same types
    int score;
    std::string alignedPattern;
    std::string alignedReference;
    std::string cigar;
    std::string mdz;
};

// hw2.cpp:118 / 192: exact signatures.  The result is heap-allocated and owned by the caller (`delete`, hw2.cpp:395-399).
// Where the reference cannot fail, these abort with a message on stderr and exit code 2 when no MI355X is usable or the
// engine reports an error (there is no CPU fallback).
AlignmentResult* globalAlignmentNeedlemanWunsch(const std::string& patterns, const std::string& references, int matchScore,
                                                int mismatchScore, int gapPenalty);
AlignmentResult* localAlignmentSmithWaterman(const std::string& patterns, const std::string& references, int matchScore,
                                             int mismatchScore, int gapPenalty);
// optional: HIP device ordinal used by the calling thread's context (default 0); call before the first alignment
void hw2DropinSetDevice(int device);
