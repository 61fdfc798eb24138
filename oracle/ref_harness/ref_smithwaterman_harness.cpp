// oracle/ref_harness/ref_smithwaterman_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Thin C entry points over the reference's own Smith-Waterman (compiled in place by oracle/Makefile
// `ref` from deepmutect/Mutect2Cpp-master/src/intel/smithwaterman/avx2_impl.cc and
// smithwaterman_common.cc): what IntelSmithWaterman.cc:58-68 `SmithWaterman_align` does, minus the
// CPU dispatch (the AVX2 and AVX-512 builds are the same template, PairWiseSW.h).
#include <cstdint>
#include <cstring>
#include <vector>

#include "avx2_impl.h"

extern "C" {

// one pair; cigar must hold cap bytes (zero-filled here, as IntelSmithWaterman::align does)
int ref_sw_align(int match, int mismatch, int open, int extend, const uint8_t* seq1, int len1, const uint8_t* seq2, int len2,
                 int strategy, char* cigar, int cap, uint32_t* count, int32_t* offset) {
    memset(cigar, 0, (size_t)cap);
    return runSWOnePairBT_fp_avx2(match, mismatch, open, extend, const_cast<uint8_t*>(seq1), const_cast<uint8_t*>(seq2),
                                  (int16_t)len1, (int16_t)len2, (int8_t)strategy, cigar, cap, count, offset);
}

// batch over concatenated sequences; cigars[p * stride ...], stride >= 2 * max(len1, len2) + 1
int ref_sw_batch(int match, int mismatch, int open, int extend, int n_pairs, const uint64_t* off1, const uint8_t* seq1,
                 const uint64_t* off2, const uint8_t* seq2, const uint8_t* strategy, char* cigars, int stride, int32_t* offsets) {
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 4)
    for (int p = 0; p < n_pairs; ++p) {
        const int l1 = (int)(off1[p + 1] - off1[p]), l2 = (int)(off2[p + 1] - off2[p]);
        uint32_t count = 0;
        char* c = cigars + (size_t)p * stride;
        memset(c, 0, (size_t)stride);
        const int cap = 2 * (l1 > l2 ? l1 : l2);                    // smithwaterman/IntelSmithWaterman.cpp:8
        if (runSWOnePairBT_fp_avx2(match, mismatch, open, extend, const_cast<uint8_t*>(seq1 + off1[p]), const_cast<uint8_t*>(seq2 + off2[p]),
                                   (int16_t)l1, (int16_t)l2, (int8_t)strategy[p], c, cap < stride ? cap : stride - 1, &count, &offsets[p]))
            bad = 1;
    }
    return bad;
}

}
