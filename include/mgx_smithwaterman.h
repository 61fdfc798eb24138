/*
 * mgx_smithwaterman.h -- C ABI of the MI355X Smith-Waterman aligner with back-trace (libmgx.so).
 *
 * the realignment kernel Mutect2Cpp runs right after PairHMM (read -> best
 * haplotype, Mutect2Engine.cpp:226-230) and for haplotype -> reference during assembly.  Reference
 * interface replaced (paths relative to deepmutect/Mutect2Cpp-master/src/):
 *
 *   mgx_sw_align        <- SmithWaterman_align(ref, refLength, alt, altLength, cigar, cigarLength,
 *                                              match, mismatch, open, extend, strategy) -> offset
 *                          intel/smithwaterman/IntelSmithWaterman.cc:58-68, which calls
 *                          runSWOnePairBT_fp_avx2 / _avx512 (intel/smithwaterman/PairWiseSW.h:447-503)
 *   mgx_sw_align_batch  <- the same for many pairs at once: the per-read loop around
 *                          SWNativeAlignerWrapper::align (smithwaterman/SWNativeAlignerWrapper.cpp:9-30)
 *   mgx_sw_create       <- smithwaterman_initial() (IntelSmithWaterman.cc:34-56, CPU dispatch there)
 *
 * Semantics are the reference's to the byte: the affine-gap recurrence and its tie rules
 * (PairWiseSW.h:31-66), the choice of the best end cell among equal scores in anti-diagonal order
 * (:256-285), the four overhang strategies, the back-trace state machine and the CIGAR text with
 * its capacity rule (:299-445; an element whose text would not fit the buffer is skipped).
 * The shortcut of SWNativeAlignerWrapper (alt found verbatim in ref -> "<altLength>M") stays host
 * code in the caller.
 *
 * Sequences are compared byte by byte (no base decoding), lengths are 1..2048 for ref and 1..32767
 * for alt; longer references are rejected with -E2BIG.  All functions return 0 (mgx_sw_align: the
 * alignment offset) or a negative errno-style code with mgx_last_error() set.  No CPU fallback.
 */
#ifndef MGX_SMITHWATERMAN_H
#define MGX_SMITHWATERMAN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mgx_sw mgx_sw_t;

/* overhang strategies, numbered as IntelSmithWaterman::getStrategy passes them down
 * (smithwaterman/IntelSmithWaterman.cpp:18-35, intel/smithwaterman/smithwaterman_common.h:53-56) */
#define MGX_SW_SOFTCLIP      9
#define MGX_SW_INDEL         10
#define MGX_SW_LEADING_INDEL 11
#define MGX_SW_IGNORE        12

/* SWParameters (smithwaterman/SWParameters.h): e.g. STANDARD_NGS = {25, -50, -110, -6},
 * ORIGINAL_DEFAULT = {3, -1, -4, -3} (smithwaterman/SmithWatermanAligner.cpp:8-9) */
typedef struct mgx_sw_params {
    int32_t match, mismatch, gap_open, gap_extend;
} mgx_sw_params_t;

typedef struct mgx_sw_input {
    uint64_t n_pairs;
    const uint64_t* ref_off;   /* [n_pairs + 1] offsets into ref (seq1: rows of the matrix) */
    const uint8_t* ref;
    const uint64_t* alt_off;   /* [n_pairs + 1] offsets into alt (seq2: columns) */
    const uint8_t* alt;
    const uint8_t* strategy;   /* [n_pairs] MGX_SW_* */
} mgx_sw_input_t;

typedef struct mgx_sw_stats {
    uint64_t n_pairs, cells;         /* cells = sum of refLength * altLength */
    uint32_t n_launches;             /* fill-kernel launches of the last batch */
    float ms_fill, ms_trace;         /* HIP events around the matrix fill and the back-trace kernels */
    uint64_t backtrace_bytes;        /* back-trace arena written by the fill kernels */
    uint64_t n_pairs_i16;            /* pairs whose scores provably fit 16 bits: filled two to a lane group with packed arithmetic */
} mgx_sw_stats_t;

int mgx_sw_create(int device, unsigned flags, mgx_sw_t** out);
void mgx_sw_destroy(mgx_sw_t* ctx);

/* Batch.  out_offset[p] = alignment offset.  out_cigar + p * cigar_stride receives the NUL-terminated
 * CIGAR text of pair p; like the reference's buffer its capacity for text is
 * min(2 * max(refLength, altLength), cigar_stride - 1) bytes.  out_score (may be NULL) = best score. */
int mgx_sw_align_batch(mgx_sw_t* ctx, const mgx_sw_params_t* params, const mgx_sw_input_t* in,
                       int32_t* out_offset, char* out_cigar, uint32_t cigar_stride, int32_t* out_score);

/* One pair, argument for argument SmithWaterman_align (IntelSmithWaterman.cc:58): writes the CIGAR text
 * into cigar (cigarLength bytes, zero-filled first; the text capacity is cigarLength, as there) and returns
 * the alignment offset -- which may be negative for IGNORE but is bounded by the 16-bit lengths -- or
 * MGX_SW_ALIGN_ERROR(errno) on failure. */
#define MGX_SW_ALIGN_ERROR(e) (-1000000 - (e))
#define MGX_SW_ALIGN_FAILED(rc) ((rc) <= -1000000)
int mgx_sw_align(mgx_sw_t* ctx, const uint8_t* ref, int refLength, const uint8_t* alt, int altLength,
                 uint8_t* cigar, int cigarLength, int match, int mismatch, int open, int extend, uint8_t strategy);

int mgx_sw_stats(mgx_sw_t* ctx, mgx_sw_stats_t* out);

#ifdef __cplusplus
}
#endif
#endif /* MGX_SMITHWATERMAN_H */
