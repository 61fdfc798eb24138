// mgx_native_shim.h -- the reference's ex-JNI "native" PairHMM functions on top of libmgx.so.
//
// Drop-in for deepmutect/Mutect2Cpp-master/src/intel/pairhmm/IntelPairHmm.cc: the functions that
// intel/pairhmm/IntelPairHmm.h:37-50 declares -- initNative, computeLikelihoodsNative,
// computeLikelihoodsNative_concurrent(_i), computeLikelihoodsNative_concurrent_trie(_i) -- keep their
// signatures over the reference's own types (`testcase`, `trie_testcase`, `ReadForPairHMM`,
// intel/pairhmm/pairhmm_common.h:45-68) and are implemented by flattening the test cases into the packed arrays
// of include/mgx_pairhmm.h and ONE mgx_pairhmm_compute call.  Compile this header inside the reference tree (its
// include paths must resolve "intel/pairhmm/pairhmm_common.h") in place of IntelPairHmm.cc:
//
//     // intel/pairhmm/IntelPairHmm.cc, whole file:
//     #define MGX_NATIVE_SHIM_IMPLEMENTATION
//     #include "mgx_native_shim.h"
//
// and link libmgx.so.  Without MGX_NATIVE_SHIM_IMPLEMENTATION only the mgx_native:: helpers are declared.
//
// What is flattened (nothing else is read):
//   read table   one entry per DISTINCT ReadForPairHMM object (VectorLoglessPairHMM de-duplicates equal reads before it
//                builds test cases, VectorLoglessPairHMM.cpp:71-104, so pointer identity is the right key): bases =
//                rs[0, rslen) (borrowed from the SAMRecord, ReadForPairHMM.cpp:19); charCombination is four consecutive
//                blocks of rslen bytes  del | ins | gcp | qual  (ReadForPairHMM.cpp:28-33), already masked with 127 (:34-36);
//   hap table    one entry per distinct (hap pointer, haplen) -- HaplotypeDataHolder borrows Haplotype::getBases();
//   test cases   (read index, haplotype index) in the order of the vector; likelihoodArray[i] belongs to testcases[i].
// The seven probability vectors ReadForPairHMM::initializeFloatVector() precomputes for the AVX kernels are not used:
// the device derives its per-row constants from the same bytes and the same Context<> tables.
//
// Semantics kept: float first, results below MIN_ACCEPTED recomputed in double, log10 minus the initial constant
// (IntelPairHmm.cc:338-350) -- on the device; use_double -> MGX_PAIRHMM_FORCE_DOUBLE (:205); flush-to-zero (:230) is a
// build flag of the device code; one context per calling thread (the reference's worker threads each own a
// VectorLoglessPairHMM, Mutect2Engine.cpp:27-29, and share only process globals, which here are two flags).
// Errors: the reference's functions return void and swallow `const char*` throws (:290-292); a device failure is not
// something to swallow, so the shim throws std::runtime_error(mgx_last_error()), which threadFunc catches per region
// (main.cpp:303-310).
#ifndef MGX_NATIVE_SHIM_H
#define MGX_NATIVE_SHIM_H

#include <atomic>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "intel/pairhmm/pairhmm_common.h"   // the reference's testcase / trie_testcase / ReadForPairHMM / ConvertChar
#include "mgx_pairhmm.h"

namespace mgx_native {

// the process-global switches of the reference (IntelPairHmm.cc:45-47: g_use_double, g_max_threads)
inline std::atomic<int>& use_double_flag() { static std::atomic<int> f{0}; return f; }
inline std::atomic<int>& device_ordinal() { static std::atomic<int> d{MGX_DEVICE_AUTO}; return d; }

struct ThreadContext {
    mgx_pairhmm_t* ctx = nullptr;
    int made_with_double = -1;
    ~ThreadContext() { if (ctx) mgx_pairhmm_destroy(ctx); }
    mgx_pairhmm_t* get() {
        const int want = use_double_flag().load();
        if (ctx && made_with_double != want) { mgx_pairhmm_destroy(ctx); ctx = nullptr; }
        if (!ctx) {
            if (mgx_pairhmm_create(device_ordinal().load(), want ? MGX_PAIRHMM_FORCE_DOUBLE : 0u, &ctx))
                throw std::runtime_error(std::string("mgx_pairhmm_create: ") + mgx_last_error());
            made_with_double = want;
        }
        return ctx;
    }
};
inline ThreadContext& thread_context() { static thread_local ThreadContext t; return t; }

// Packed form of a list of test cases; owns the arrays `in` points into.
struct Flattened {
    std::vector<uint64_t> read_off{0}, hap_off{0};
    std::vector<uint8_t> bases, qual, ins, del, gcp, hap_bases;
    std::vector<uint32_t> pair_read, pair_hap;
    std::unordered_map<const ReadForPairHMM*, uint32_t> read_index;
    struct HapKey {
        const void* p; uint64_t len;
        bool operator==(const HapKey& o) const { return p == o.p && len == o.len; }
    };
    struct HapHash { size_t operator()(const HapKey& k) const { return std::hash<const void*>()(k.p) ^ (size_t)(k.len * 0x9E3779B97F4A7C15ull); } };
    std::unordered_map<HapKey, uint32_t, HapHash> hap_index;
    mgx_pairhmm_input_t in{};

    uint32_t add_read(const ReadForPairHMM* r) {
        auto it = read_index.find(r);
        if (it != read_index.end()) return it->second;
        if (r->rslen <= 0) throw std::invalid_argument("mgx_native: read of length 0");
        const size_t n = (size_t)r->rslen;
        const uint8_t* cc = reinterpret_cast<const uint8_t*>(r->charCombination);     // del | ins | gcp | qual, ReadForPairHMM.cpp:28-33
        bases.insert(bases.end(), r->rs, r->rs + n);
        del.insert(del.end(), cc, cc + n);
        ins.insert(ins.end(), cc + n, cc + 2 * n);
        gcp.insert(gcp.end(), cc + 2 * n, cc + 3 * n);
        qual.insert(qual.end(), cc + 3 * n, cc + 4 * n);
        read_off.push_back(bases.size());
        const uint32_t idx = (uint32_t)read_index.size();
        read_index.emplace(r, idx);
        return idx;
    }
    uint32_t add_hap(const void* p, uint64_t len) {
        HapKey k{p, len};
        auto it = hap_index.find(k);
        if (it != hap_index.end()) return it->second;
        const uint8_t* b = static_cast<const uint8_t*>(p);
        hap_bases.insert(hap_bases.end(), b, b + len);
        hap_off.push_back(hap_bases.size());
        const uint32_t idx = (uint32_t)hap_index.size();
        hap_index.emplace(k, idx);
        return idx;
    }
    void add_case(const ReadForPairHMM* r, const void* hap, uint64_t haplen) {
        pair_read.push_back(add_read(r));
        pair_hap.push_back(add_hap(hap, haplen));
    }
    const mgx_pairhmm_input_t* finish() {
        in.n_reads = read_off.size() - 1; in.read_off = read_off.data();
        in.bases = bases.data(); in.qual = qual.data(); in.ins = ins.data(); in.del = del.data(); in.gcp = gcp.data();
        in.n_haps = hap_off.size() - 1; in.hap_off = hap_off.data(); in.hap_bases = hap_bases.data();
        in.n_pairs = pair_read.size(); in.pair_read = pair_read.data(); in.pair_hap = pair_hap.data();
        return &in;
    }
};

// test cases [begin, end) -> out[begin, end)
inline void compute_range(std::vector<testcase>& tcs, std::vector<double>& out, size_t begin, size_t end) {
    if (begin >= end) return;
    if (out.size() < end) throw std::invalid_argument("mgx_native: likelihoodArray is smaller than the test case list");
    Flattened f;
    f.pair_read.reserve(end - begin); f.pair_hap.reserve(end - begin);
    for (size_t i = begin; i < end; ++i) f.add_case(tcs[i].readForPairHmm.get(), tcs[i].hap, (uint64_t)tcs[i].haplen);
    if (mgx_pairhmm_compute(thread_context().get(), f.finish(), out.data() + begin))
        throw std::runtime_error(std::string("mgx_pairhmm_compute: ") + mgx_last_error());
}

// every trie test case = one read against ALL haplotypes of its haplotypeDataArray, results appended in that order
// (IntelPairHmm.cc:695-708; the trie itself is a CPU-only saving, SURVEY.md A13, and is not walked)
inline void compute_trie_range(std::vector<trie_testcase>& tcs, std::vector<std::vector<double>>& out, size_t begin, size_t end) {
    if (begin >= end) return;
    if (out.size() < end) throw std::invalid_argument("mgx_native: likelihoodArray is smaller than the test case list");
    Flattened f;
    size_t total = 0;
    for (size_t i = begin; i < end; ++i) total += tcs[i].haplotypeDataArray.size();
    f.pair_read.reserve(total); f.pair_hap.reserve(total);
    for (size_t i = begin; i < end; ++i)
        for (const HaplotypeDataHolder& h : tcs[i].haplotypeDataArray)
            f.add_case(tcs[i].readForPairHmm.get(), h.haplotypeBases, h.length);
    std::vector<double> flat(total);
    if (total && mgx_pairhmm_compute(thread_context().get(), f.finish(), flat.data()))
        throw std::runtime_error(std::string("mgx_pairhmm_compute: ") + mgx_last_error());
    size_t at = 0;
    for (size_t i = begin; i < end; ++i) {
        const size_t nh = tcs[i].haplotypeDataArray.size();
        out[i].insert(out[i].end(), flat.begin() + at, flat.begin() + at + nh);     // emplace_back per haplotype, :702-705
        at += nh;
    }
}

}  // namespace mgx_native

#ifdef MGX_NATIVE_SHIM_IMPLEMENTATION
// ---- the functions intel/pairhmm/IntelPairHmm.h:37-50 declares -----------------------------------------------------
void initNative(bool use_double, int /*max_threads: OpenMP threads of the CPU kernels; the device needs none*/) {
    mgx_native::use_double_flag().store(use_double ? 1 : 0);       // IntelPairHmm.cc:205
    ConvertChar::init();                                           // :254 (other reference code may read the table)
    mgx_native::thread_context().get();                            // fail here, not in the first region, when no device is visible
}

void computeLikelihoodsNative(std::vector<testcase>& testcases, std::vector<double>& likelihoodArray) {
    mgx_native::compute_range(testcases, likelihoodArray, 0, testcases.size());
}

// The reference walks the list one test case at a time and, in the tail phase, shares the rest with idle worker threads
// (IntelPairHmm.cc:296-330).  One device batch replaces both the walk and the sharing.
void computeLikelihoodsNative_concurrent(std::vector<testcase>& testcases, std::vector<double>& likelihoodArray) {
    mgx_native::compute_range(testcases, likelihoodArray, 0, testcases.size());
}

// One test case per call: correct, but a device round trip per test case -- callers that loop over i
// (VectorLoglessPairHMM.cpp:118-119) should make ONE computeLikelihoodsNative call instead.
void computeLikelihoodsNative_concurrent_i(std::vector<testcase>& testcases, std::vector<double>& likelihoodArray, unsigned long i) {
    mgx_native::compute_range(testcases, likelihoodArray, i, i + 1);
}

void computeLikelihoodsNative_concurrent_trie(std::vector<trie_testcase>& testcases, std::vector<std::vector<double>>& likelihoodArray) {
    mgx_native::compute_trie_range(testcases, likelihoodArray, 0, testcases.size());
}

void computeLikelihoodsNative_concurrent_trie_i(std::vector<trie_testcase>& testcases, std::vector<std::vector<double>>& likelihoodArray,
                                                unsigned long i) {
    mgx_native::compute_trie_range(testcases, likelihoodArray, i, i + 1);
}
#endif  // MGX_NATIVE_SHIM_IMPLEMENTATION

#endif  // MGX_NATIVE_SHIM_H
