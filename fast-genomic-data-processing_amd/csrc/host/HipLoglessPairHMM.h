// HipLoglessPairHMM.h -- host-side C++ adapter that puts the MI355X engine behind the reference's
// PairHMM class boundary (deepmutect/Mutect2Cpp-master/src/utils/pairhmm/PairHMM.h:13-69).
//
// It does what VectorLoglessPairHMM does around the native call
// (utils/pairhmm/VectorLoglessPairHMM.cpp:18-41, 43-148) and nothing more:
//   initialize()               haplotype table + haplotype -> index map; is_use_trietree_optimize
//                              is always false (the trie is a CPU-only saving with identical
//                              results)
//   enqueue() / flush()        the same for several active regions in one device batch (row F1)
//   computeLog10Likelihoods()  per read: the four quality arrays (&127 like ReadForPairHMM.cpp:34-36),
//                              de-duplication of identical reads (same bases and same four arrays),
//                              one test case per (unique read, haplotype), ONE batched call into
//                              the C ABI instead of the per-test-case loop (:118-119), scatter to
//                              logLikelihoods->set(alleleIdx, readIdx, value) honouring the allele
//                              order of the matrix (:135-146)
//
// The class is a template over the reference's own types so that it compiles both inside the
// reference tree (`using HipPairHMM = mgx::HipLoglessPairHMM<RefTraits>;`)
// and stand-alone against mock types (tests/cpp/test_adapter.cpp).  A Traits type provides:
//
//   using Haplotype, Read, Matrix, GcpMap;
//   static const uint8_t* hap_bases(const Haplotype&);   static int hap_len(const Haplotype&);
//   static int read_len(const Read&);
//   static const uint8_t* read_bases(const Read&);       static const uint8_t* read_quals(const Read&);
//   static std::shared_ptr<uint8_t[]> ins_quals(const std::shared_ptr<Read>&, int len);   // ReadUtils::getBaseInsertionQualities
//   static std::shared_ptr<uint8_t[]> del_quals(const std::shared_ptr<Read>&, int len);   // ReadUtils::getBaseDeletionQualities
//   static const char* gcp(GcpMap&, Read*);
//   static <iterable of shared_ptr<Haplotype>> alleles(Matrix&);   static void set(Matrix&, int allele, int read, double v);
#pragma once

#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "mgx_pairhmm.h"

namespace mgx {

template <class Traits>
class HipLoglessPairHMM {
public:
    using Haplotype = typename Traits::Haplotype;
    using Read = typename Traits::Read;
    using Matrix = typename Traits::Matrix;
    using GcpMap = typename Traits::GcpMap;

    // mirrors VectorLoglessPairHMM(PairHMMNativeArgumentCollection&): initNative(useDoublePrecision, threads)
    explicit HipLoglessPairHMM(bool use_double_precision = false, int device = 0) {
        if (mgx_pairhmm_create(device, use_double_precision ? MGX_PAIRHMM_FORCE_DOUBLE : 0u, &ctx_) != 0)
            throw std::runtime_error(std::string("mgx_pairhmm_create: ") + mgx_last_error());
    }
    ~HipLoglessPairHMM() { mgx_pairhmm_destroy(ctx_); }
    HipLoglessPairHMM(const HipLoglessPairHMM&) = delete;
    HipLoglessPairHMM& operator=(const HipLoglessPairHMM&) = delete;

    bool is_use_trietree_optimize = false;       // read by PairHMMLikelihoodCalculationEngine.cpp:69

    // PairHMM::initialize (the per-sample read list is only used for sizing in the reference)
    template <class PerSampleReads>
    void initialize(const std::vector<std::shared_ptr<Haplotype>>& haplotypes, const PerSampleReads&,
                    int /*readMaxLength*/, int /*haplotypeMaxLength*/) {
        hap_off_.assign(1, 0);
        hap_bases_.clear();
        hap_index_.clear();
        for (const auto& h : haplotypes) {
            const int len = Traits::hap_len(*h);
            const uint8_t* b = Traits::hap_bases(*h);
            hap_bases_.insert(hap_bases_.end(), b, b + len);
            hap_off_.push_back(hap_bases_.size());
            hap_index_.emplace(h.get(), (int)hap_index_.size());
        }
        is_use_trietree_optimize = false;
    }

    void computeLog10Likelihoods(Matrix* logLikelihoods, std::vector<std::shared_ptr<Read>>& processedReads,
                                 GcpMap* gcp) {
        if (processedReads.empty()) return;
        Prepared p = prepare(logLikelihoods, processedReads, gcp);
        mgx_pairhmm_input_t in = p.input();
        if (mgx_pairhmm_compute(ctx_, &in, p.out.data()) != 0)
            throw std::runtime_error(std::string("mgx_pairhmm_compute: ") + mgx_last_error());
        scatter(p);
    }

    // SURVEY.md 8f row F1 -- region-level batching inside the caller: enqueue() does everything
    // computeLog10Likelihoods does up to the native call (for the haplotypes of the latest initialize())
    // and parks the region; flush() sends every parked region to the device as ONE batch
    // (mgx_pairhmm_compute_regions) and fills the matrices.  A worker can therefore assemble the next
    // regions while earlier ones wait, and small regions no longer pay a device round trip each
    // (1000 regions of 40 x 25: 163 us per region one at a time, 29 us in one batch).
    // The matrices must stay alive until flush().
    void enqueue(Matrix* logLikelihoods, std::vector<std::shared_ptr<Read>>& processedReads, GcpMap* gcp) {
        if (processedReads.empty()) return;
        queue_.push_back(prepare(logLikelihoods, processedReads, gcp));
    }
    size_t queued() const { return queue_.size(); }
    void flush() {
        if (queue_.empty()) return;
        std::vector<mgx_pairhmm_input_t> ins;
        std::vector<double*> outs;
        for (auto& p : queue_) { ins.push_back(p.input()); outs.push_back(p.out.data()); }
        const int rc = mgx_pairhmm_compute_regions(ctx_, (uint32_t)ins.size(), ins.data(), outs.data());
        if (rc != 0) { queue_.clear(); throw std::runtime_error(std::string("mgx_pairhmm_compute_regions: ") + mgx_last_error()); }
        for (auto& p : queue_) scatter(p);
        queue_.clear();
    }

    // The same through a host work queue shared by the worker threads (mgx_pairhmm_queue_*, BASELINE configs[2]): the
    // parked regions are cut into batches that the queue's lanes -- possibly on several GPUs -- pull, pack and upload
    // while earlier batches compute.  Values are identical to flush().
    void flush(mgx_pairhmm_queue_t* queue) {
        if (queue_.empty()) return;
        std::vector<mgx_pairhmm_input_t> ins;
        std::vector<double*> outs;
        for (auto& p : queue_) { ins.push_back(p.input()); outs.push_back(p.out.data()); }
        const int rc = mgx_pairhmm_queue_run_regions(queue, (uint32_t)ins.size(), ins.data(), outs.data());
        if (rc != 0) { queue_.clear(); throw std::runtime_error(std::string("mgx_pairhmm_queue_run_regions: ") + mgx_last_error()); }
        for (auto& p : queue_) scatter(p);
        queue_.clear();
    }

    // the reference calls the _trie variants only when is_use_trietree_optimize is true; they are
    // provided so the virtual interface is complete and return the same values
    void computeLog10Likelihoods_trie(Matrix* m, std::vector<std::shared_ptr<Read>>& r, GcpMap* g) { computeLog10Likelihoods(m, r, g); }
    void computeLog10Likelihoods_trie_unique(Matrix* m, std::vector<std::shared_ptr<Read>>& r, GcpMap* g) { computeLog10Likelihoods(m, r, g); }

private:
    // one region, ready for the native call: unique reads x the haplotype table of its initialize()
    struct Prepared {
        Matrix* matrix = nullptr;
        int n_reads = 0, n_haps = 0;
        std::vector<uint64_t> read_off{0}, hap_off;
        std::vector<uint8_t> bases, qual, ins, del, gc, hap_bases;
        std::vector<int> unique_of;
        std::unordered_map<const Haplotype*, int> hap_index;
        std::vector<double> out;
        mgx_pairhmm_input_t input() const {
            mgx_pairhmm_input_t in{};
            in.n_reads = read_off.size() - 1; in.read_off = read_off.data();
            in.bases = bases.data(); in.qual = qual.data(); in.ins = ins.data(); in.del = del.data(); in.gcp = gc.data();
            in.n_haps = (uint64_t)n_haps; in.hap_off = hap_off.data(); in.hap_bases = hap_bases.data();
            in.n_pairs = in.n_reads * in.n_haps; in.pair_read = nullptr; in.pair_hap = nullptr;
            return in;
        }
    };

    Prepared prepare(Matrix* logLikelihoods, std::vector<std::shared_ptr<Read>>& processedReads, GcpMap* gcp) {
        Prepared p;
        p.matrix = logLikelihoods;
        p.n_reads = (int)processedReads.size();
        p.n_haps = (int)hap_off_.size() - 1;
        p.hap_off = hap_off_; p.hap_bases = hap_bases_; p.hap_index = hap_index_;
        // ---- unique reads (VectorLoglessPairHMM.cpp:71-104): key = quals (masked) + bases
        p.unique_of.resize(p.n_reads);
        std::unordered_map<std::string, int> seen;
        for (int r = 0; r < p.n_reads; ++r) {
            const auto& rd = processedReads[r];
            const int len = Traits::read_len(*rd);
            auto iq = Traits::ins_quals(rd, len);
            auto dq = Traits::del_quals(rd, len);
            const uint8_t* q = Traits::read_quals(*rd);
            const uint8_t* b = Traits::read_bases(*rd);
            const char* g = Traits::gcp(*gcp, rd.get());
            std::string key;
            key.resize((size_t)5 * len);
            for (int k = 0; k < len; ++k) {
                key[k] = (char)(dq[k] & 127);            key[len + k] = (char)(iq[k] & 127);
                key[2 * len + k] = (char)(g[k] & 127);   key[3 * len + k] = (char)(q[k] & 127);
                key[4 * len + k] = (char)b[k];
            }
            auto it = seen.find(key);
            if (it != seen.end()) { p.unique_of[r] = it->second; continue; }
            const int u = (int)p.read_off.size() - 1;
            seen.emplace(std::move(key), u);
            p.unique_of[r] = u;
            p.bases.insert(p.bases.end(), b, b + len);
            p.qual.insert(p.qual.end(), q, q + len);
            p.ins.insert(p.ins.end(), iq.get(), iq.get() + len);
            p.del.insert(p.del.end(), dq.get(), dq.get() + len);
            p.gc.insert(p.gc.end(), (const uint8_t*)g, (const uint8_t*)g + len);
            p.read_off.push_back(p.bases.size());
        }
        // one test case per (unique read, haplotype), read-major like uniqueTestcases: the
        // cross-product form of the ABI (pair arrays NULL), out[u * n_haps + h]
        p.out.resize((p.read_off.size() - 1) * (size_t)p.n_haps);
        return p;
    }

    // VectorLoglessPairHMM.cpp:135-146
    static void scatter(Prepared& p) {
        for (int r = 0; r < p.n_reads; ++r) {
            int hapIdx = 0;
            for (auto& haplotype : Traits::alleles(*p.matrix)) {
                const int idx = p.hap_index.at(haplotype.get());
                Traits::set(*p.matrix, hapIdx, r, p.out[(size_t)p.unique_of[r] * p.n_haps + idx]);
                hapIdx++;
            }
        }
    }

    std::vector<Prepared> queue_;
    mgx_pairhmm_t* ctx_ = nullptr;
    std::vector<uint64_t> hap_off_{0};
    std::vector<uint8_t> hap_bases_;
    std::unordered_map<const Haplotype*, int> hap_index_;
};

}  // namespace mgx
