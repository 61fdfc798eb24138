// oracle/ref_harness/ref_pairhmm_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Thin driver around the reference's OWN PairHMM translation units, which are compiled where
// they lie under /root/reference by oracle/Makefile (nothing is copied into this repo).  It
// exists only in the build container: oracle/_ref/libref_pairhmm.so travels to the GPU box as
// a prebuilt checker / CPU baseline, the reference sources never do.
//
// Reference TUs linked: intel/pairhmm/{avx_impl.cc, avx512_impl.cc, pairhmm_common.cc},
// haplotypecaller/ReadForPairHMM.cpp, trie/trieNode.cpp.
// NOT linked: intel/pairhmm/IntelPairHmm.cc -- it includes boost/utility.hpp and boost is not
// in this image, so that TU is unbuildable here.  Its role on this path is the twelve lines at
// IntelPairHmm.cc:202-256 (kernel selection, FTZ, ConvertChar::init) and :332-351 (float first,
// < MIN_ACCEPTED -> double, log10 - LOG10_INITIAL_CONSTANT); those are restated below against
// the reference's own symbols (compute_fp_*, Context<>, MIN_ACCEPTED, ConvertChar).
#include <cstdint>
#include <cmath>
#include <memory>
#include <vector>
#include <xmmintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "intel/common/avx.h"
#include "intel/pairhmm/pairhmm_common.h"
#include "intel/pairhmm/Context.h"
#include "intel/pairhmm/avx_impl.h"
#include "intel/pairhmm/avx512_impl.h"

static Context<float> g_ctxf;    // IntelPairHmm.cc:48-49: table init happens in the ctor
static Context<double> g_ctxd;
static float (*g_f)(testcase*) = nullptr;
static double (*g_d)(testcase*) = nullptr;

extern "C" int ref_pairhmm_init() {
    // IntelPairHmm.cc:233-251
    if (is_avx512_supported()) { g_f = compute_fp_avx512s; g_d = compute_fp_avx512d; }
    else                       { g_f = compute_fp_avxs;    g_d = compute_fp_avxd; }
    ConvertChar::init();         // IntelPairHmm.cc:254
    return is_avx512_supported() ? 512 : 256;
}

extern "C" int ref_pairhmm_batch(int64_t n_pairs, const uint64_t* read_off, const uint8_t* bases,
                                 const uint8_t* qual, const uint8_t* ins, const uint8_t* del,
                                 const uint8_t* gcp, const uint64_t* hap_off,
                                 const uint8_t* hap_bases, const uint32_t* pair_read,
                                 const uint32_t* pair_hap, double* out_log10,
                                 uint8_t* used_double, int threads) {
    if (!g_f) ref_pairhmm_init();
    int used = 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    used = threads;
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
#endif
    for (int64_t i = 0; i < n_pairs; i++) {
        _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);   // IntelPairHmm.cc:230 (per thread)
        uint64_t ro = read_off[pair_read[i]], ho = hap_off[pair_hap[i]];
        int R = (int)(read_off[pair_read[i] + 1] - ro);
        int H = (int)(hap_off[pair_hap[i] + 1] - ho);
        // VectorLoglessPairHMM.cpp:80-87
        auto read = std::make_shared<ReadForPairHMM>(R, qual + ro, ins + ro, del + ro,
                                                     (const char*)(gcp + ro), bases + ro);
        read->initializeFloatVector();
        testcase tc(H, hap_bases + ho, read);
        // IntelPairHmm.cc:338-350
        double result_final;
        float result_float = g_f(&tc);
        if (result_float < MIN_ACCEPTED) {
            double result_double = g_d(&tc);
            result_final = log10(result_double) - Context<double>::LOG10_INITIAL_CONSTANT;
            if (used_double) used_double[i] = 1;
        } else {
            result_final = (double)(log10f(result_float) - Context<float>::LOG10_INITIAL_CONSTANT);
            if (used_double) used_double[i] = 0;
        }
        out_log10[i] = result_final;
    }
    return used;
}
