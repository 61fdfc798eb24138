// mgx_pairhmm.hip -- host side of the PairHMM C ABI (include/mgx_pairhmm.h) for MI355X.
//
// Replaces, behind a plain C boundary, the reference's native PairHMM layer
// (deepmutect/Mutect2Cpp-master/src/intel/pairhmm/IntelPairHmm.cc:202-351): table set-up,
// float-first / double-fallback policy and the per-test-case loop.  The per-test-case loop
// becomes: bin test cases by read-length class, sort each bin by haplotype length, launch one
// fp32 kernel per bin, then one fp64 kernel per bin over the device-side re-run list.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <vector>

#include "../../include/mgx_pairhmm.h"
#include "mgx_common.h"
#include "mgx_tables.h"
#include "pairhmm_pack.h"
#include "pairhmm_kernels.hip.inc"

using mgx::set_error;

namespace {

// Row classes: a read of R rows is owned by G lanes with RPL = ceil(R / G) rows each.  Few rows per lane waste
// issue slots on the per-step overhead (3 DPP moves, the haplotype byte, loop control are paid per step, not per
// cell), many lanes waste steps on fill/drain (G - 1 per pair) and lanes on padding (up to G - 1 rows), so short
// reads take narrow groups: G = 4 up to 32 bases, G = 8 up to 64, G = 16 up to 192 (1..12 rows per lane: the
// common 100-151 bp reads), G = 64 with 4..16 rows per lane beyond that (up to 1024 bases).
// (Giving 8 lanes up to 16 rows each -- reads up to 128 bases -- was measured and lost: 128 x 256 test cases 6019 ->
// 5642 GCUPS, ragged 4429 -> 3835; profiles/r02_pairhmm_g8_rows.txt.)
constexpr int kNarrowMaxRPL = 8, kG8MaxRPL = 8, kG16MaxRPL = 12, kG64MinRPL = 4, kG64MaxRPL = 16;
constexpr int kMaxRowsG4 = 4 * kNarrowMaxRPL, kMaxRowsG8 = 8 * kG8MaxRPL;
constexpr int kMaxRowsG16 = 16 * kG16MaxRPL;
constexpr int kMaxRowsG64 = 64 * kG64MaxRPL;
// Longer reads are strip-mined: 64 lanes x 16 rows per strip, the strips sweep the haplotype one after the other
// (pairhmm_fwd_strip).  The limit below only keeps row indices and the boundary scratch in 32 bits.
constexpr int kMaxRowsStrip = 1 << 20;
constexpr int kStripMarkRPL = kG64MaxRPL + 1;      // shape_of's RPL value for the strip-mined class
constexpr uint32_t kMaxLdsPerBlock = 64 * 1024;

struct Bin {
    int G = 0, RPL = 0;              // fp32 kernel shape
    int Gd = 0, RPLd = 0;            // fp64 re-run kernel shape (fewer registers per row class)
    uint32_t job_begin = 0, job_count = 0;
    uint32_t max_h = 0;
    uint64_t cells = 0, alg_bytes = 0;
    // launch geometry
    uint32_t block = 256, lds_stride = 0, grid_f32 = 0, grid_f64 = 0, grid_f64_all = 0;
    bool strip = false;              // reads longer than 1024 bases: the strip-mined kernel, one test case per workgroup
    bool pk = false;                 // fp32 launches use the packed kernel (pairhmm_fwd_pk<G, RPL / 2>): even RPL <= 8, G <= 16
};

}  // namespace

namespace {
// One device allocation + (for batches up to kStageLimit) a pinned host mirror of the same size.
// Slabs are recycled through the context, so a stream of region-sized batches does no hipMalloc.
struct Slab {
    uint8_t* dev = nullptr;
    uint8_t* pin = nullptr;     // mirrors the first pin_cap bytes only: inputs and results, never device scratch
    size_t cap = 0, pin_cap = 0;
};
constexpr size_t kStageLimit = 256u << 20;
constexpr size_t kMaxEventSets = 64;       // runs whose kernel durations a timed batch remembers
constexpr size_t kMinSlab = 1u << 20;
}  // namespace

struct mgx_pairhmm {
    int device = 0;
    unsigned flags = 0;
    hipStream_t compute = nullptr, copy = nullptr, d2h = nullptr;   // kernels | uploads | result downloads
    // a batch with several read-length classes launches one kernel per class: they are dealt to the compute
    // stream and these side streams so that the drain of one launch overlaps the body of the others
    static constexpr int kAux = 3;
    hipStream_t aux[kAux] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kAux] = {nullptr, nullptr, nullptr};
    int n_streams = 1;
    float* d_ph2pr_f = nullptr; float* d_mm_f = nullptr; float* d_div3_f = nullptr; float* d_ratio_f = nullptr;
    double* d_ph2pr_d = nullptr; double* d_mm_d = nullptr; double* d_div3_d = nullptr; double* d_ratio_d = nullptr;
    float log10_initial_f = 0; double log10_initial_d = 0;
    int n_cu = 256;
    std::vector<Slab> free_slabs;
    void* d_strip = nullptr; size_t strip_cap = 0;      // boundary rows of the strip-mined class (one compute stream: launches do not overlap)
};

struct mgx_pairhmm_batch {
    uint64_t n_pairs = 0;
    std::vector<Bin> bins;
    // everything lives in one slab: [jobs | bases | qual | ins | del | gcp | hap] is written once
    // (one H2D copy when staged through the pinned mirror), [out | used] is read back with one
    // D2H copy, [rerun_list | rerun_count] is device scratch
    Slab slab;
    size_t in_bytes = 0, o_out = 0, o_used = 0, result_bytes = 0;
    uint8_t *d_bases = nullptr, *d_qual = nullptr, *d_ins = nullptr, *d_del = nullptr,
            *d_gcp = nullptr, *d_hap = nullptr, *d_used = nullptr;
    Job* d_jobs = nullptr;
    uint32_t* d_rerun_list = nullptr;
    uint32_t* d_rerun_count = nullptr;
    double* d_out = nullptr;
    std::vector<Job> host_jobs;    // only kept for unstaged (very large) batches
    // whole-region form (mgx_pairhmm_region): normalise / filter epilogue over [n_reads][n_haps]
    bool has_model = false;
    uint32_t n_reads = 0, n_haps = 0;
    uint64_t* d_read_len = nullptr;
    uint8_t* d_keep = nullptr;
    uint32_t *d_row_off = nullptr, *d_row_nh = nullptr;   // several regions in one batch: per-read output row
    size_t o_keep = 0;
    double log10_rate = 0, max_err = 0;
    hipEvent_t uploaded = nullptr;
    // timing: a ring of event sets, one set per run (4 per bin: f32 start/stop, f64 start/stop), so that
    // every run of a timed loop is measured and batch_stats can average them after the final sync
    std::vector<hipEvent_t> ev;
    // what every class's fp32 event pair covers in the last run: its own launch, a whole multi-class launch (booked on
    // the member with the most cells) or nothing (the other members)
    std::vector<uint64_t> acct_cells, acct_bytes;
    std::vector<int8_t> acct_multi;    // 0: single-class kernel, 1 + GSET: multi-class kernel
    uint32_t ev_sets = 0;          // sets allocated
    uint32_t runs_timed = 0;       // runs recorded since the last batch_stats
    hipEvent_t done = nullptr;     // recorded on the compute stream behind the last kernel of a run
    bool ran = false;
    mgx_pairhmm_stats_t stats{};
};

namespace {

using BatchPtr = std::unique_ptr<mgx_pairhmm_batch, void (*)(mgx_pairhmm_batch*)>;
BatchPtr new_batch() {
    // (on an error path the slab is freed, not pooled: destroy is called without the context)
    return BatchPtr(new (std::nothrow) mgx_pairhmm_batch, [](mgx_pairhmm_batch* p) { mgx_pairhmm_batch_destroy(nullptr, p); });
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,     \
                      __LINE__);                                                           \
            return -EIO;                                                                   \
        }                                                                                  \
    } while (0)

template <typename T>
int upload(T** dst, const void* src, size_t bytes, hipStream_t s) {
    *dst = nullptr;
    if (bytes == 0) bytes = 16;
    HIP_TRY(hipMalloc((void**)dst, bytes));
    if (src) HIP_TRY(hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, s));
    return 0;
}

// (G, RPL) class of a read of R rows; G = 0 if unsupported.  MGX_PAIRHMM_MIN_G=16 switches the narrow groups off.
inline void shape_of(uint32_t R, int* G, int* RPL) {
    static const int min_g = [] { const char* e = getenv("MGX_PAIRHMM_MIN_G"); const int v = e ? atoi(e) : 4; return v; }();
    const uint32_t g8_rows = (uint32_t)kMaxRowsG8;
    if (R <= (uint32_t)kMaxRowsG4 && min_g <= 4) { *G = 4; *RPL = (int)((R + 3) / 4); }
    else if (R <= g8_rows && min_g <= 8) { *G = 8; *RPL = (int)((R + 7) / 8); }
    else if (R <= (uint32_t)kMaxRowsG16) { *G = 16; *RPL = (int)((R + 15) / 16); }
    else if (R <= (uint32_t)kMaxRowsG64) { *G = 64; *RPL = std::max(kG64MinRPL, (int)((R + 63) / 64)); }
    else if (R <= (uint32_t)kMaxRowsStrip) { *G = 64; *RPL = kStripMarkRPL; }
    else { *G = 0; *RPL = 0; }
}
// dynamic LDS of one block: per-wavefront emission table (fp32 only) + per-group haplotype codes
inline uint32_t lds_bytes(const Bin& bin, bool f32) {
    const uint32_t waves = bin.block / 64u, groups = bin.block / (uint32_t)(f32 ? bin.G : bin.Gd);
    uint32_t etab = f32 ? (uint32_t)((bin.RPL + 1) / 2) * kNumCodes * 512u : 0u;
    if (f32 && bin.pk) etab = (uint32_t)((bin.RPL * kPkRowStride + 15) & ~15);
    return waves * etab + groups * bin.lds_stride;
}
// bins in order of (G, RPL): [G=4: 1..8][G=8: 1..8][G=16: 1..12][G=64: 4..16]
constexpr int kBinG8 = kNarrowMaxRPL, kBinG16 = kBinG8 + kG8MaxRPL, kBinG64 = kBinG16 + kG16MaxRPL;
constexpr int kBinStrip = kBinG64 + (kG64MaxRPL - kG64MinRPL + 1);      // the strip-mined class comes last
constexpr int kBins = kBinStrip + 1;
inline int bin_index(int G, int RPL) {
    if (G == 64 && RPL == kStripMarkRPL) return kBinStrip;
    return G == 4 ? RPL - 1 : G == 8 ? kBinG8 + RPL - 1 : G == 16 ? kBinG16 + RPL - 1 : kBinG64 + RPL - kG64MinRPL;
}
inline void bin_shape(int k, Bin* b) {
    if (k == kBinStrip) { b->G = 64; b->RPL = kG64MaxRPL; b->Gd = 64; b->RPLd = kG64MaxRPL; b->strip = true; return; }
    if (k < kBinG8) { b->G = 4; b->RPL = k + 1; }
    else if (k < kBinG16) { b->G = 8; b->RPL = k - kBinG8 + 1; }
    else if (k < kBinG64) { b->G = 16; b->RPL = k - kBinG16 + 1; }
    else { b->G = 64; b->RPL = k - kBinG64 + kG64MinRPL; }
    // the fp64 kernel keeps twice the registers per row: beyond 8 rows per lane of 16 it runs one
    // pair per wavefront instead
    if (b->G == 16 && b->RPL > 8) { b->Gd = 64; b->RPLd = (16 * b->RPL + 63) / 64; }
    else { b->Gd = b->G; b->RPLd = b->RPL; }
}
constexpr uint64_t kMergeBelow = 4096;

// launch geometry of a bin once job_count and max_h are known
int finalize_bin(Bin& bin, int n_cu, unsigned ctx_flags) {
    // LDS per group: G pad + codes + G pad + prefetch slack (see the kernel's staging loop)
    // the packed fp32 kernel is opt-in: context flag MGX_PAIRHMM_PACKED_FP32 or MGX_PAIRHMM_PK=1
    static const int pk_mode = [] { const char* e = getenv("MGX_PAIRHMM_PK"); return e ? atoi(e) : 0; }();
    bin.pk = (pk_mode != 0 || (ctx_flags & MGX_PAIRHMM_PACKED_FP32)) && !bin.strip && bin.G <= 16 && bin.RPL % 2 == 0 && bin.RPL <= 8;
    // (the packed kernel's half-lanes need 2G pad codes on either side of the haplotype, the scalar one G)
    bin.lds_stride = (bin.max_h + (bin.pk ? 4u : 2u) * (uint32_t)std::max(bin.G, bin.Gd) + 8u + 15u) & ~15u;
    bin.block = 64;                   // one wavefront per workgroup: finest LDS/VGPR packing per CU, no cross-wave
                                      // barriers; in-process A/B at 1 M pairs: 5.48 ms (64), 5.65 (128), 5.46 (256)
    if (const char* e = getenv("MGX_PAIRHMM_BLOCK")) { const int v = atoi(e); if (v == 64 || v == 128 || v == 256) bin.block = (uint32_t)v; }
    while (bin.block > 64u && lds_bytes(bin, true) > kMaxLdsPerBlock) bin.block /= 2;
    if (lds_bytes(bin, true) > 160u * 1024u) {
        set_error("haplotype of %u bases does not fit the LDS staging buffer", bin.max_h);
        return -E2BIG;
    }
    if (bin.strip) {
        // one test case per workgroup, a bounded grid walking the job list: the boundary rows between strips live in
        // a scratch array indexed by workgroup
        bin.block = 64;
        bin.grid_f32 = bin.grid_f64 = bin.grid_f64_all = std::min<uint32_t>(bin.job_count, (uint32_t)n_cu * 4u);
        return 0;
    }
    const uint32_t gpb = bin.block / bin.G, gpbd = bin.block / bin.Gd;
    bin.grid_f32 = (bin.job_count + gpb - 1) / gpb;
    bin.grid_f64_all = (bin.job_count + gpbd - 1) / gpbd;
    bin.grid_f64 = std::min<uint32_t>(bin.grid_f64_all, (uint32_t)n_cu * 8u);
    return 0;
}
// fold sparsely populated bins into the next larger row class of the same group width
void merge_small_bins(uint64_t (&count)[kBins], int (&remap)[kBins]) {
    for (int k = 0; k < kBins; ++k) remap[k] = k;
    for (int k = 0; k + 1 < kBins; ++k) {
        if (k + 1 == kBinG8 || k + 1 == kBinG16 || k + 1 == kBinG64 || k + 1 == kBinStrip) continue;   // never across a group-width boundary
        if (count[k] && count[k] < kMergeBelow) { count[k + 1] += count[k]; count[k] = 0; remap[k] = k + 1; }
    }
    for (int k = 0; k < kBins; ++k) { int t = k; while (remap[t] != t) t = remap[t]; remap[k] = t; }
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

int acquire_slab(mgx_pairhmm* c, size_t bytes, size_t pin_bytes, Slab* out) {
    int best = -1;
    for (size_t i = 0; i < c->free_slabs.size(); ++i) {
        const Slab& f = c->free_slabs[i];
        if (f.cap >= bytes && f.pin_cap >= pin_bytes && (best < 0 || f.cap < c->free_slabs[best].cap)) best = (int)i;
    }
    if (best >= 0) { *out = c->free_slabs[best]; c->free_slabs.erase(c->free_slabs.begin() + best); return 0; }
    auto round = [](size_t want) {
        size_t cap = kMinSlab;
        while (cap < want) cap *= 2;
        if (cap > want + (want >> 2) && want > (64u << 20)) cap = align_up(want, 1u << 20);   // no 2x waste on big batches
        return cap;
    };
    Slab sl;
    sl.cap = round(bytes);
    HIP_TRY(hipMalloc((void**)&sl.dev, sl.cap));
    if (pin_bytes) {
        sl.pin_cap = std::min(sl.cap, round(pin_bytes));
        if (hipHostMalloc((void**)&sl.pin, sl.pin_cap, hipHostMallocDefault) != hipSuccess) {
            (void)hipFree(sl.dev);
            set_error("hipHostMalloc of %zu bytes failed", sl.pin_cap);
            return -ENOMEM;
        }
    }
    *out = sl;
    return 0;
}
void release_slab(mgx_pairhmm* c, Slab sl) {
    if (!sl.dev) return;
    if (c && c->free_slabs.size() < 16) { c->free_slabs.push_back(sl); return; }
    (void)hipFree(sl.dev);
    if (sl.pin) (void)hipHostFree(sl.pin);
}

using mgx::validate;

}  // namespace

extern "C" {

int mgx_pairhmm_table_f32(int which, const float** out) {
    const auto& t = mgx::tables<float>();
    if (which == 0) { *out = t.ph2pr.data(); return mgx::kPh2prSize; }
    if (which == 1) { *out = t.mm.data(); return mgx::kMmSize; }
    return -EINVAL;
}
int mgx_pairhmm_table_f64(int which, const double** out) {
    const auto& t = mgx::tables<double>();
    if (which == 0) { *out = t.ph2pr.data(); return mgx::kPh2prSize; }
    if (which == 1) { *out = t.mm.data(); return mgx::kMmSize; }
    return -EINVAL;
}

int mgx_pairhmm_create(int device, unsigned flags, mgx_pairhmm_t** out) {
    if (!out) { set_error("out is NULL"); return -EINVAL; }
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        set_error("no HIP device is visible (this library has no CPU fallback)");
        return -ENODEV;
    }
    if (device == -1) {                 // MGX_DEVICE_AUTO: contexts are dealt round-robin over the visible GPUs
        static std::atomic<unsigned> next{0};
        device = (int)(next.fetch_add(1) % (unsigned)n_dev);
    }
    if (device < 0 || device >= n_dev) { set_error("device %d out of range (0..%d)", device, n_dev - 1); return -EINVAL; }
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<mgx_pairhmm> c(new (std::nothrow) mgx_pairhmm);
    if (!c) return -ENOMEM;
    c->device = device;
    c->flags = flags;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount;
    {
        // optional CU partition (flags bits 8..15: an 8-bit pattern repeated over the CU index): lets a
        // VALU-bound PairHMM context and an HBM-bound sort context share one GPU side by side instead
        // of queueing behind each other (BASELINE.json configs[4])
        const unsigned pat = (flags >> 8) & 0xFFu;
        if (pat != 0 && pat != 0xFFu) {
            const int n_words = (c->n_cu + 31) / 32;
            std::vector<uint32_t> mask(n_words, 0);
            for (int cu = 0; cu < c->n_cu; ++cu) if ((pat >> (cu & 7)) & 1u) mask[cu >> 5] |= 1u << (cu & 31);
            HIP_TRY(hipExtStreamCreateWithCUMask(&c->compute, (uint32_t)n_words, mask.data()));
        } else {
            HIP_TRY(hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking));
        }
    }
    HIP_TRY(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->d2h, hipStreamNonBlocking));
    {
        const unsigned pat = (flags >> 8) & 0xFFu;
        const char* e = getenv("MGX_PAIRHMM_STREAMS");
        // default 1: measured on an MI355X (profiles/r02_pairhmm_streams_ab.txt) two streams gain 2 % on resident ragged
        // batches (4529 -> 4610 GCUPS) but lose 20 % for region batches through a 4-lane queue (3089 -> 2431: lanes x
        // streams oversubscribe the device); three and four streams lose everywhere (ragged 3865 / 3895)
        c->n_streams = e ? std::max(1, std::min(1 + mgx_pairhmm::kAux, atoi(e))) : 1;
        if (pat != 0 && pat != 0xFFu) c->n_streams = 1;          // a CU-masked context keeps to its masked stream
        for (int a = 0; a + 1 < c->n_streams; ++a) {
            HIP_TRY(hipStreamCreateWithFlags(&c->aux[a], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&c->ev_join[a], hipEventDisableTiming));
        }
        HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    }
    const auto& tf = mgx::tables<float>();
    const auto& td = mgx::tables<double>();
    int rc;
    if ((rc = upload(&c->d_ph2pr_f, tf.ph2pr.data(), tf.ph2pr.size() * 4, c->compute))) return rc;
    if ((rc = upload(&c->d_mm_f, tf.mm.data(), tf.mm.size() * 4, c->compute))) return rc;
    if ((rc = upload(&c->d_ph2pr_d, td.ph2pr.data(), td.ph2pr.size() * 8, c->compute))) return rc;
    if ((rc = upload(&c->d_mm_d, td.mm.data(), td.mm.size() * 8, c->compute))) return rc;
    if ((rc = upload(&c->d_div3_f, tf.ph2pr_div3.data(), tf.ph2pr_div3.size() * 4, c->compute))) return rc;
    if ((rc = upload(&c->d_ratio_f, tf.gap_ratio.data(), tf.gap_ratio.size() * 4, c->compute))) return rc;
    if ((rc = upload(&c->d_div3_d, td.ph2pr_div3.data(), td.ph2pr_div3.size() * 8, c->compute))) return rc;
    if ((rc = upload(&c->d_ratio_d, td.gap_ratio.data(), td.gap_ratio.size() * 8, c->compute))) return rc;
    c->log10_initial_f = tf.log10_initial;
    c->log10_initial_d = td.log10_initial;
    HIP_TRY(hipStreamSynchronize(c->compute));
    *out = c.release();
    return 0;
}

void mgx_pairhmm_destroy(mgx_pairhmm_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipFree(c->d_ph2pr_f); (void)hipFree(c->d_mm_f);
    (void)hipFree(c->d_ph2pr_d); (void)hipFree(c->d_mm_d);
    (void)hipFree(c->d_div3_f); (void)hipFree(c->d_ratio_f); (void)hipFree(c->d_div3_d); (void)hipFree(c->d_ratio_d);
    for (auto& sl : c->free_slabs) { (void)hipFree(sl.dev); if (sl.pin) (void)hipHostFree(sl.pin); }
    (void)hipFree(c->d_strip);
    if (c->compute) (void)hipStreamDestroy(c->compute);
    if (c->copy) (void)hipStreamDestroy(c->copy);
    if (c->d2h) (void)hipStreamDestroy(c->d2h);
    for (int a = 0; a < mgx_pairhmm::kAux; ++a) { if (c->aux[a]) (void)hipStreamDestroy(c->aux[a]); if (c->ev_join[a]) (void)hipEventDestroy(c->ev_join[a]); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    delete c;
}

void mgx_pairhmm_batch_destroy(mgx_pairhmm_t* c, mgx_pairhmm_batch_t* b) {
    if (!b) return;
    if (c) {
        (void)hipSetDevice(c->device);
        // the slab goes back to the pool: nothing may still be reading or writing it.  Only THIS batch's
        // work is waited for (its last run, or its upload if it never ran; result downloads are
        // synchronous), so retiring one batch does not stall on the batches queued behind it.
        if (b->done) (void)hipEventSynchronize(b->done);
        else if (b->uploaded) (void)hipEventSynchronize(b->uploaded);
    }
    release_slab(c, b->slab);
    if (b->uploaded) (void)hipEventDestroy(b->uploaded);
    if (b->done) (void)hipEventDestroy(b->done);
    for (auto e : b->ev) (void)hipEventDestroy(e);
    delete b;
}

namespace {

// Batch of every read against every haplotype (pair_read == pair_hap == NULL):
// out[r * n_haps + h].  Host work is O(n_reads + n_haps); job descriptors are made on the device.
int create_cross(mgx_pairhmm_t* c, const mgx_pairhmm_input_t* in, mgx_pairhmm_batch* b,
                 const uint8_t* mapq = nullptr, const mgx_read_model_t* model = nullptr) {
    const uint64_t nr = in->n_reads, nh = in->n_haps;
    const uint64_t n = nr * nh;
    if (n > 0xFFFFFFF0ull) { set_error("more than 2^32 test cases in one batch"); return -E2BIG; }
    b->n_pairs = n;
    b->stats.n_pairs = n;
    int rc;
    uint64_t count[kBins] = {0};
    std::vector<uint8_t> rbin(nr);
    uint64_t sumR_bin[kBins] = {0}, sumH = 0;
    uint32_t max_h = 0;
    for (uint64_t r = 0; r < nr; ++r) {
        const uint64_t R = in->read_off[r + 1] - in->read_off[r];
        if (R == 0) { set_error("read %llu is empty", (unsigned long long)r); return -EINVAL; }
        int G, RPL;
        shape_of((uint32_t)std::min<uint64_t>(R, 0xFFFFFFFFull), &G, &RPL);
        if (G == 0) { set_error("read %llu: read of %llu bases exceeds the %d-row limit", (unsigned long long)r, (unsigned long long)R, kMaxRowsStrip); return -E2BIG; }
        rbin[r] = (uint8_t)bin_index(G, RPL);
        count[rbin[r]] += nh;
    }
    std::vector<SeqRef> haps(nh);
    for (uint64_t h = 0; h < nh; ++h) {
        const uint64_t H = in->hap_off[h + 1] - in->hap_off[h];
        if (H == 0) { set_error("haplotype %llu is empty", (unsigned long long)h); return -EINVAL; }
        if (H > 0x7FFFFFF0ull) { set_error("haplotype too long"); return -E2BIG; }
        haps[h] = SeqRef{in->hap_off[h], (uint32_t)H, (uint32_t)h};
        sumH += H; max_h = std::max<uint32_t>(max_h, (uint32_t)H);
    }
    std::stable_sort(haps.begin(), haps.end(), [](const SeqRef& a, const SeqRef& b2) { return a.len > b2.len; });
    int remap[kBins];
    merge_small_bins(count, remap);
    uint64_t reads_in[kBins] = {0}, rstart[kBins + 1] = {0};
    for (uint64_t r = 0; r < nr; ++r) { rbin[r] = (uint8_t)remap[rbin[r]]; reads_in[rbin[r]]++; sumR_bin[rbin[r]] += in->read_off[r + 1] - in->read_off[r]; }
    for (int k = 0; k < kBins; ++k) rstart[k + 1] = rstart[k] + reads_in[k];
    // slab layout
    const uint64_t read_bytes = in->read_off[nr], hap_bytes = in->hap_off[nh];
    size_t off = 0;
    const size_t o_rtab = off;  off = align_up(off + nr * sizeof(SeqRef));
    const size_t o_htab = off;  off = align_up(off + nh * sizeof(SeqRef));
    const size_t o_bases = off; off = align_up(off + read_bytes);
    const size_t o_qual = off;  off = align_up(off + read_bytes);
    const size_t o_ins = off;   off = align_up(off + read_bytes);
    const size_t o_del = off;   off = align_up(off + read_bytes);
    const size_t o_gcp = off;   off = align_up(off + read_bytes);
    const size_t o_hap = off;   off = align_up(off + hap_bytes);
    const size_t o_mapq = off;  off = align_up(off + (model ? nr : 0));
    const size_t o_rlen = off;  off = align_up(off + (model ? nr * sizeof(uint64_t) : 0));
    b->in_bytes = off;
    b->o_out = off;             off = align_up(off + n * sizeof(double));
    b->o_used = off;            off = align_up(off + n);
    b->o_keep = off;            off = align_up(off + (model ? nr : 0));
    b->result_bytes = off - b->o_out;
    const size_t pin_bytes = off;              // the pinned mirror covers the inputs and the results only
    const size_t o_jobs = off;  off = align_up(off + n * sizeof(Job));
    const size_t o_rlist = off; off = align_up(off + 2 * n * sizeof(uint32_t));   // fp64 re-run lists | exact-tier list
    const size_t o_rcount = off; off = align_up(off + 64 * sizeof(uint32_t));
    if ((rc = acquire_slab(c, off, pin_bytes, &b->slab))) return rc;
    uint8_t* dv = b->slab.dev; uint8_t* pin = b->slab.pin;
    b->d_jobs = (Job*)(dv + o_jobs);
    b->d_bases = dv + o_bases; b->d_qual = dv + o_qual; b->d_ins = dv + o_ins; b->d_del = dv + o_del;
    b->d_gcp = dv + o_gcp; b->d_hap = dv + o_hap;
    b->d_out = (double*)(dv + b->o_out); b->d_used = dv + b->o_used;
    b->d_rerun_list = (uint32_t*)(dv + o_rlist); b->d_rerun_count = (uint32_t*)(dv + o_rcount);
    SeqRef* rtab = (SeqRef*)(pin + o_rtab);
    {
        uint64_t cur[kBins];
        for (int k = 0; k < kBins; ++k) cur[k] = rstart[k];
        for (uint64_t r = 0; r < nr; ++r)
            rtab[cur[rbin[r]]++] = SeqRef{in->read_off[r], (uint32_t)(in->read_off[r + 1] - in->read_off[r]), (uint32_t)r};
    }
    memcpy(pin + o_htab, haps.data(), nh * sizeof(SeqRef));
    memcpy(pin + o_bases, in->bases, read_bytes); memcpy(pin + o_qual, in->qual, read_bytes);
    memcpy(pin + o_ins, in->ins, read_bytes);     memcpy(pin + o_del, in->del, read_bytes);
    memcpy(pin + o_gcp, in->gcp, read_bytes);     memcpy(pin + o_hap, in->hap_bases, hap_bytes);
    hipStream_t s = c->copy;
    if (model) {
        for (uint64_t r = 0; r < nr; ++r) {
            ((uint64_t*)(pin + o_rlen))[r] = in->read_off[r + 1] - in->read_off[r];
        }
        memcpy(pin + o_mapq, mapq, nr);
        b->has_model = true; b->n_reads = (uint32_t)nr; b->n_haps = (uint32_t)nh;
        b->d_read_len = (uint64_t*)(dv + o_rlen); b->d_keep = dv + b->o_keep;
        b->log10_rate = model->log10_mismapping_rate; b->max_err = model->max_error_per_base;
    }
    HIP_TRY(hipMemcpyAsync(dv, pin, b->in_bytes, hipMemcpyHostToDevice, s));
    if (model) {
        // modifyReadQualities + gap continuation penalties, in place on the uploaded arrays
        ReadModel rm{};
        rm.rate_factor = model->pcr_rate_factor; rm.bq_threshold = model->base_quality_threshold;
        rm.constant_gcp = model->constant_gcp;
        for (int i = 0; i <= 20; ++i) {      // PairHMMLikelihoodCalculationEngine.cpp:45-61
            const double d = 40.0 - std::exp((double)i / ((double)std::max(rm.rate_factor, 1) * M_PI));
            const int v = (d > 0.0 ? (int)(d + 0.5) : (int)(d - 0.5)) + 1;
            rm.pcr_cache[i] = (uint8_t)(char)std::max(10, v);
        }
        hipLaunchKernelGGL(pairhmm_read_model, dim3((uint32_t)nr), dim3(128), 0, s, (const SeqRef*)(dv + o_rtab),
                           b->d_bases, b->d_qual, b->d_ins, b->d_del, b->d_gcp, (const uint8_t*)(dv + o_mapq), rm);
    }
    uint64_t job_begin = 0;
    for (int k = 0; k < kBins; ++k) {
        if (!reads_in[k]) continue;
        Bin bin;
        bin_shape(k, &bin);
        bin.job_begin = (uint32_t)job_begin;
        bin.job_count = (uint32_t)(reads_in[k] * nh);
        bin.max_h = max_h;
        bin.cells = sumR_bin[k] * sumH;
        bin.alg_bytes = 5 * sumR_bin[k] * nh + reads_in[k] * (sumH + 4 * nh);
        if ((rc = finalize_bin(bin, c->n_cu, c->flags))) return rc;
        hipLaunchKernelGGL(pairhmm_make_jobs, dim3((bin.job_count + 255) / 256), dim3(256), 0, s,
                           (const SeqRef*)(dv + o_rtab) + rstart[k], (const SeqRef*)(dv + o_htab), (uint32_t)nh,
                           bin.job_count, b->d_jobs + job_begin);
        b->stats.cells += bin.cells; b->stats.alg_bytes += bin.alg_bytes;
        b->bins.push_back(bin);
        job_begin += bin.job_count;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventCreateWithFlags(&b->uploaded, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(b->uploaded, s));
    return 0;
}

// Row F1: the cross-product test cases of several regions in one batch -- one upload, one set of
// launches, one download.  Host work is linear in reads + haplotypes (the test cases are enumerated
// on the device), so coalescing a thousand small regions costs microseconds, not a pair list.
int create_cross_multi(mgx_pairhmm_t* c, uint32_t n_regions, const mgx_pairhmm_input_t* regs, mgx_pairhmm_batch* b,
                       std::vector<uint64_t>* out_base, const uint8_t* const* mapq = nullptr, const mgx_read_model_t* model = nullptr) {
    uint64_t nr = 0, nh = 0, n = 0, read_bytes = 0, hap_bytes = 0;
    out_base->assign(n_regions + 1, 0);
    for (uint32_t g = 0; g < n_regions; ++g) {
        const mgx_pairhmm_input_t& in = regs[g];
        if (in.pair_read || in.pair_hap) { set_error("region %u: regions are given in the cross-product form (pair arrays NULL)", g); return -EINVAL; }
        nr += in.n_reads; nh += in.n_haps; n += in.n_reads * in.n_haps;
        if (in.n_reads) read_bytes += in.read_off[in.n_reads] - in.read_off[0];
        if (in.n_haps) hap_bytes += in.hap_off[in.n_haps] - in.hap_off[0];
        (*out_base)[g + 1] = n;
    }
    if (n > 0xFFFFFFF0ull || nr > 0xFFFFFFF0ull) { set_error("more than 2^32 test cases in one batch"); return -E2BIG; }
    b->n_pairs = n;
    b->stats.n_pairs = n;
    if (n == 0) return 0;
    int rc;
    // global read table: bin, length, region; job counts per bin
    struct RInfo { uint64_t off; uint32_t len, region; uint8_t bin; };
    std::vector<RInfo> rinfo(nr);
    std::vector<RegionRef> rtab_regions(n_regions);
    std::vector<SeqRef> haps(nh);
    uint64_t count[kBins] = {0};
    uint32_t max_h = 0;
    {
        uint64_t r_at = 0, h_at = 0, rb = 0, hb = 0;
        for (uint32_t g = 0; g < n_regions; ++g) {
            const mgx_pairhmm_input_t& in = regs[g];
            RegionRef& R = rtab_regions[g];
            R.hap_begin = (uint32_t)h_at; R.n_haps = (uint32_t)in.n_haps; R.read_base = (uint32_t)r_at; R.out_base = (uint32_t)(*out_base)[g];
            for (uint64_t h = 0; h < in.n_haps; ++h) {
                const uint64_t H = in.hap_off[h + 1] - in.hap_off[h];
                if (H == 0) { set_error("region %u: haplotype %llu is empty", g, (unsigned long long)h); return -EINVAL; }
                if (H > 0x7FFFFFF0ull) { set_error("haplotype too long"); return -E2BIG; }
                haps[h_at + h] = SeqRef{hb + (in.hap_off[h] - in.hap_off[0]), (uint32_t)H, (uint32_t)h};
                max_h = std::max<uint32_t>(max_h, (uint32_t)H);
            }
            std::stable_sort(haps.begin() + h_at, haps.begin() + h_at + in.n_haps, [](const SeqRef& a, const SeqRef& b2) { return a.len > b2.len; });
            for (uint64_t r = 0; r < in.n_reads; ++r) {
                const uint64_t Rl = in.read_off[r + 1] - in.read_off[r];
                if (Rl == 0) { set_error("region %u: read %llu is empty", g, (unsigned long long)r); return -EINVAL; }
                int G, RPL;
                shape_of((uint32_t)std::min<uint64_t>(Rl, 0xFFFFFFFFull), &G, &RPL);
                if (G == 0) { set_error("region %u: read of %llu bases exceeds the %d-row limit", g, (unsigned long long)Rl, kMaxRowsStrip); return -E2BIG; }
                RInfo& ri = rinfo[r_at + r];
                ri.off = rb + (in.read_off[r] - in.read_off[0]); ri.len = (uint32_t)Rl; ri.region = g; ri.bin = (uint8_t)bin_index(G, RPL);
                count[ri.bin] += in.n_haps;
            }
            if (in.n_reads) rb += in.read_off[in.n_reads] - in.read_off[0];
            if (in.n_haps) hb += in.hap_off[in.n_haps] - in.hap_off[0];
            r_at += in.n_reads; h_at += in.n_haps;
        }
    }
    int remap[kBins];
    merge_small_bins(count, remap);
    uint64_t reads_in[kBins] = {0}, rstart[kBins + 1] = {0}, jobs_in[kBins] = {0}, cells_in[kBins] = {0}, bytes_in[kBins] = {0};
    std::vector<uint64_t> sumH(n_regions, 0);
    for (uint32_t g = 0; g < n_regions; ++g)
        for (uint32_t h = 0; h < rtab_regions[g].n_haps; ++h) sumH[g] += haps[rtab_regions[g].hap_begin + h].len;
    for (uint64_t r = 0; r < nr; ++r) {
        RInfo& ri = rinfo[r];
        ri.bin = (uint8_t)remap[ri.bin];
        const RegionRef& R = rtab_regions[ri.region];
        reads_in[ri.bin]++; jobs_in[ri.bin] += R.n_haps; cells_in[ri.bin] += (uint64_t)ri.len * sumH[ri.region];
        bytes_in[ri.bin] += 5ull * ri.len * R.n_haps + sumH[ri.region] + 4ull * R.n_haps;
    }
    for (int k = 0; k < kBins; ++k) rstart[k + 1] = rstart[k] + reads_in[k];
    // slab layout
    size_t off = 0;
    const size_t o_rtab = off;  off = align_up(off + nr * sizeof(SeqRef));
    const size_t o_rreg = off;  off = align_up(off + nr * sizeof(uint32_t));
    const size_t o_rpre = off;  off = align_up(off + (nr + 1) * sizeof(uint32_t));
    const size_t o_gtab = off;  off = align_up(off + n_regions * sizeof(RegionRef));
    const size_t o_htab = off;  off = align_up(off + nh * sizeof(SeqRef));
    const size_t o_bases = off; off = align_up(off + read_bytes);
    const size_t o_qual = off;  off = align_up(off + read_bytes);
    const size_t o_ins = off;   off = align_up(off + read_bytes);
    const size_t o_del = off;   off = align_up(off + read_bytes);
    const size_t o_gcp = off;   off = align_up(off + read_bytes);
    const size_t o_hap = off;   off = align_up(off + hap_bytes);
    const size_t o_mapq = off;  off = align_up(off + (model ? nr : 0));
    const size_t o_rlen = off;  off = align_up(off + (model ? nr * sizeof(uint64_t) : 0));
    const size_t o_roff = off;  off = align_up(off + (model ? nr * sizeof(uint32_t) : 0));
    const size_t o_rnh = off;   off = align_up(off + (model ? nr * sizeof(uint32_t) : 0));
    b->in_bytes = off;
    b->o_out = off;             off = align_up(off + n * sizeof(double));
    b->o_used = off;            off = align_up(off + n);
    b->o_keep = off;            off = align_up(off + (model ? nr : 0));
    b->result_bytes = off - b->o_out;
    const size_t pin_bytes = off;              // the pinned mirror covers the inputs and the results only
    const size_t o_jobs = off;  off = align_up(off + n * sizeof(Job));
    const size_t o_rlist = off; off = align_up(off + 2 * n * sizeof(uint32_t));   // fp64 re-run lists | exact-tier list
    const size_t o_rcount = off; off = align_up(off + 64 * sizeof(uint32_t));
    if ((rc = acquire_slab(c, off, pin_bytes, &b->slab))) return rc;
    uint8_t* dv = b->slab.dev; uint8_t* pin = b->slab.pin;
    b->d_jobs = (Job*)(dv + o_jobs);
    b->d_bases = dv + o_bases; b->d_qual = dv + o_qual; b->d_ins = dv + o_ins; b->d_del = dv + o_del;
    b->d_gcp = dv + o_gcp; b->d_hap = dv + o_hap;
    b->d_out = (double*)(dv + b->o_out); b->d_used = dv + b->o_used;
    b->d_rerun_list = (uint32_t*)(dv + o_rlist); b->d_rerun_count = (uint32_t*)(dv + o_rcount);
    SeqRef* rtab = (SeqRef*)(pin + o_rtab);
    uint32_t* rreg = (uint32_t*)(pin + o_rreg);
    uint32_t* rpre = (uint32_t*)(pin + o_rpre);
    {
        uint64_t cur[kBins];
        for (int k = 0; k < kBins; ++k) cur[k] = rstart[k];
        for (uint64_t r = 0; r < nr; ++r) {
            const RInfo& ri = rinfo[r];
            const uint64_t at = cur[ri.bin]++;
            rtab[at] = SeqRef{ri.off, ri.len, (uint32_t)r};
            rreg[at] = ri.region;
        }
        uint32_t run = 0;
        for (uint64_t at = 0; at < nr; ++at) { rpre[at] = run; run += rtab_regions[rreg[at]].n_haps; }
        rpre[nr] = run;
    }
    memcpy(pin + o_gtab, rtab_regions.data(), n_regions * sizeof(RegionRef));
    memcpy(pin + o_htab, haps.data(), nh * sizeof(SeqRef));
    {
        size_t rb = 0, hb = 0;
        for (uint32_t g = 0; g < n_regions; ++g) {
            const mgx_pairhmm_input_t& in = regs[g];
            if (in.n_reads) {
                const uint64_t o0 = in.read_off[0], len = in.read_off[in.n_reads] - o0;
                memcpy(pin + o_bases + rb, in.bases + o0, len); memcpy(pin + o_qual + rb, in.qual + o0, len);
                memcpy(pin + o_ins + rb, in.ins + o0, len);     memcpy(pin + o_del + rb, in.del + o0, len);
                memcpy(pin + o_gcp + rb, in.gcp + o0, len);
                rb += len;
            }
            if (in.n_haps) {
                const uint64_t o0 = in.hap_off[0], len = in.hap_off[in.n_haps] - o0;
                memcpy(pin + o_hap + hb, in.hap_bases + o0, len);
                hb += len;
            }
        }
    }
    hipStream_t s = c->copy;
    if (model) {
        // per read, in global read order: MAPQ, length, and its row of the output
        uint64_t* rlen = (uint64_t*)(pin + o_rlen);
        uint32_t* roff = (uint32_t*)(pin + o_roff);
        uint32_t* rnh = (uint32_t*)(pin + o_rnh);
        uint64_t r_at = 0;
        for (uint32_t g = 0; g < n_regions; ++g) {
            const mgx_pairhmm_input_t& in = regs[g];
            if (in.n_reads && !mapq[g]) { set_error("region %u: mapq is NULL", g); return -EINVAL; }
            for (uint64_t r = 0; r < in.n_reads; ++r) {
                const uint64_t len = in.read_off[r + 1] - in.read_off[r];
                rlen[r_at + r] = len;
                roff[r_at + r] = (uint32_t)((*out_base)[g] + r * in.n_haps);
                rnh[r_at + r] = (uint32_t)in.n_haps;
            }
            if (in.n_reads) memcpy(pin + o_mapq + r_at, mapq[g], in.n_reads);
            r_at += in.n_reads;
        }
        b->has_model = true; b->n_reads = (uint32_t)nr; b->n_haps = 0;
        b->d_read_len = (uint64_t*)(dv + o_rlen); b->d_keep = dv + b->o_keep;
        b->d_row_off = (uint32_t*)(dv + o_roff); b->d_row_nh = (uint32_t*)(dv + o_rnh);
        b->log10_rate = model->log10_mismapping_rate; b->max_err = model->max_error_per_base;
    }
    HIP_TRY(hipMemcpyAsync(dv, pin, b->in_bytes, hipMemcpyHostToDevice, s));
    if (model) {
        ReadModel rm{};
        rm.rate_factor = model->pcr_rate_factor; rm.bq_threshold = model->base_quality_threshold;
        rm.constant_gcp = model->constant_gcp;
        for (int i = 0; i <= 20; ++i) {      // PairHMMLikelihoodCalculationEngine.cpp:45-61
            const double d = 40.0 - std::exp((double)i / ((double)std::max(rm.rate_factor, 1) * M_PI));
            const int v = (d > 0.0 ? (int)(d + 0.5) : (int)(d - 0.5)) + 1;
            rm.pcr_cache[i] = (uint8_t)(char)std::max(10, v);
        }
        hipLaunchKernelGGL(pairhmm_read_model, dim3((uint32_t)nr), dim3(128), 0, s, (const SeqRef*)(dv + o_rtab),
                           b->d_bases, b->d_qual, b->d_ins, b->d_del, b->d_gcp, (const uint8_t*)(dv + o_mapq), rm);
    }
    uint64_t job_begin = 0;
    for (int k = 0; k < kBins; ++k) {
        if (!reads_in[k]) continue;
        Bin bin;
        bin_shape(k, &bin);
        bin.job_begin = (uint32_t)job_begin;
        bin.job_count = (uint32_t)jobs_in[k];
        bin.max_h = max_h;
        bin.cells = cells_in[k];
        bin.alg_bytes = bytes_in[k];
        if ((rc = finalize_bin(bin, c->n_cu, c->flags))) return rc;
        b->stats.cells += bin.cells; b->stats.alg_bytes += bin.alg_bytes;
        b->bins.push_back(bin);
        job_begin += bin.job_count;
    }
    // one enumeration launch for all bins: jobs are laid out in read-table order, which is bin order
    hipLaunchKernelGGL(pairhmm_make_jobs_multi, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, (const SeqRef*)(dv + o_rtab),
                       (const uint32_t*)(dv + o_rreg), (const uint32_t*)(dv + o_rpre), (uint32_t)nr, (const RegionRef*)(dv + o_gtab),
                       (const SeqRef*)(dv + o_htab), (uint32_t)n, b->d_jobs);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventCreateWithFlags(&b->uploaded, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(b->uploaded, s));
    return 0;
}

}  // namespace

namespace {

// Pair-list batch.  plan == nullptr: the test cases of `in` with its read / haplotype arrays copied as
// they are.  plan != nullptr: test cases [plan->lo, plan->hi) of `in`, only the sequences they reference
// (gathered by the packer, pairhmm_pack.h) -- the form the host work queue uploads.
int create_pairs(mgx_pairhmm_t* c, const mgx_pairhmm_input_t* in, const mgx::PackPlan* plan, mgx_pairhmm_batch* b) {
    const uint64_t n = plan ? plan->hi - plan->lo : in->n_pairs;
    const uint32_t* pr = plan ? plan->pair_read.data() : in->pair_read;
    const uint32_t* ph = plan ? plan->pair_hap.data() : in->pair_hap;
    const uint64_t* roff = plan ? plan->roff.data() : in->read_off;
    const uint64_t* hoff = plan ? plan->hoff.data() : in->hap_off;
    const uint64_t n_reads = plan ? plan->lread.size() : in->n_reads, n_haps = plan ? plan->lhap.size() : in->n_haps;
    int rc;
    b->n_pairs = n;
    b->stats.n_pairs = n;

    // ---- bin by (G, RPL), then counting-sort every bin by haplotype length -------------
    std::vector<uint32_t> bin_of(n);
    uint64_t count[kBins] = {0};
    uint32_t max_h = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t r = pr[i], h = ph[i];
        if (r >= n_reads || h >= n_haps) { set_error("test case %llu: index out of range", (unsigned long long)i); return -EINVAL; }
        const uint64_t R = roff[r + 1] - roff[r];
        const uint64_t H = hoff[h + 1] - hoff[h];
        if (R == 0 || H == 0) { set_error("test case %llu: empty read or haplotype", (unsigned long long)i); return -EINVAL; }
        int G, RPL;
        shape_of((uint32_t)std::min<uint64_t>(R, 0xFFFFFFFFull), &G, &RPL);
        if (G == 0) { set_error("test case %llu: read of %llu bases exceeds the %d-row limit", (unsigned long long)i, (unsigned long long)R, kMaxRowsStrip); return -E2BIG; }
        if (H > 0x7FFFFFF0ull) { set_error("haplotype too long"); return -E2BIG; }
        const int bi = bin_index(G, RPL);
        bin_of[i] = (uint32_t)bi;
        count[bi]++;
        max_h = std::max<uint32_t>(max_h, (uint32_t)H);
        b->stats.cells += R * H;
        b->stats.alg_bytes += 5 * R + H + 4;
    }
    // A bin with few jobs is folded into the next larger row class of the same group width (any
    // RPL >= ceil(R/G) is valid, it only leaves lanes unused): a region-sized batch then needs one
    // fp32 + one fp64 launch instead of one pair per read-length class.
    {
        int remap[kBins];
        merge_small_bins(count, remap);
        for (uint64_t i = 0; i < n; ++i) bin_of[i] = (uint32_t)remap[bin_of[i]];
    }
    // ---- one slab for everything; layout decided before the jobs are written so that they can be
    //      built directly in the pinned mirror
    const uint64_t read_bytes = roff[n_reads];
    const uint64_t hap_bytes = hoff[n_haps];
    size_t off = 0;
    const size_t o_jobs = off;  off = align_up(off + n * sizeof(Job));
    const size_t o_bases = off; off = align_up(off + read_bytes);
    const size_t o_qual = off;  off = align_up(off + read_bytes);
    const size_t o_ins = off;   off = align_up(off + read_bytes);
    const size_t o_del = off;   off = align_up(off + read_bytes);
    const size_t o_gcp = off;   off = align_up(off + read_bytes);
    const size_t o_hap = off;   off = align_up(off + hap_bytes);
    b->in_bytes = off;
    b->o_out = off;             off = align_up(off + n * sizeof(double));
    b->o_used = off;            off = align_up(off + n);
    b->result_bytes = off - b->o_out;
    const size_t o_rlist = off; off = align_up(off + 2 * n * sizeof(uint32_t));   // fp64 re-run lists | exact-tier list
    const size_t o_rcount = off; off = align_up(off + 64 * sizeof(uint32_t));
    const bool staged = plan || o_rlist <= kStageLimit;
    if ((rc = acquire_slab(c, off, staged ? o_rlist : 0, &b->slab))) return rc;
    uint8_t* dv = b->slab.dev;
    b->d_jobs = (Job*)(dv + o_jobs);
    b->d_bases = dv + o_bases; b->d_qual = dv + o_qual; b->d_ins = dv + o_ins; b->d_del = dv + o_del;
    b->d_gcp = dv + o_gcp; b->d_hap = dv + o_hap;
    b->d_out = (double*)(dv + b->o_out); b->d_used = dv + b->o_used;
    b->d_rerun_list = (uint32_t*)(dv + o_rlist); b->d_rerun_count = (uint32_t*)(dv + o_rcount);
    if (!staged) b->host_jobs.resize(n);
    Job* jobs = staged ? (Job*)(b->slab.pin + o_jobs) : b->host_jobs.data();
    {
        // key = (bin, H descending): counting sort on H inside each bin keeps the wavefront's groups
        // (consecutive jobs) at near-equal step counts, and longest first means the last workgroups of a
        // launch -- its drain -- are the shortest jobs.
        std::vector<uint64_t> bin_start(kBins + 1, 0);
        for (int k = 0; k < kBins; ++k) bin_start[k + 1] = bin_start[k] + count[k];
        std::vector<std::vector<uint32_t>> hist(kBins);
        for (int k = 0; k < kBins; ++k) if (count[k]) hist[k].assign((size_t)max_h + 2, 0);
        for (uint64_t i = 0; i < n; ++i) {
            const uint32_t h = ph[i];
            const uint32_t H = (uint32_t)(hoff[h + 1] - hoff[h]);
            hist[bin_of[i]][(max_h - H) + 1]++;
        }
        for (int k = 0; k < kBins; ++k)
            for (size_t x = 1; x < hist[k].size(); ++x) hist[k][x] += hist[k][x - 1];
        for (uint64_t i = 0; i < n; ++i) {
            const uint32_t r = pr[i], h = ph[i];
            const uint32_t R = (uint32_t)(roff[r + 1] - roff[r]);
            const uint32_t H = (uint32_t)(hoff[h + 1] - hoff[h]);
            const int k = (int)bin_of[i];
            Job& jb = jobs[bin_start[k] + hist[k][max_h - H]++];
            jb.read_off = roff[r]; jb.hap_off = hoff[h];
            jb.R = R; jb.H = H; jb.pair = (uint32_t)i; jb.pad_ = 0;
        }
        for (int k = 0; k < kBins; ++k) {
            if (!count[k]) continue;
            Bin bin;
            bin_shape(k, &bin);
            bin.job_begin = (uint32_t)bin_start[k];
            bin.job_count = (uint32_t)count[k];
            for (uint64_t q = bin_start[k]; q < bin_start[k + 1]; ++q) {
                bin.max_h = std::max(bin.max_h, jobs[q].H);
                bin.cells += (uint64_t)jobs[q].R * jobs[q].H;
                bin.alg_bytes += 5ull * jobs[q].R + jobs[q].H + 4;
            }
            if ((rc = finalize_bin(bin, c->n_cu, c->flags))) return rc;
            b->bins.push_back(bin);
        }
    }

    // ---- upload: one copy out of the pinned mirror, or (very large batches) one per array
    hipStream_t s = c->copy;
    if (staged) {
        uint8_t* pin = b->slab.pin;
        if (plan) {
            mgx::pack_copy(in, *plan, pin + o_bases, pin + o_qual, pin + o_ins, pin + o_del, pin + o_gcp, pin + o_hap);
        } else if (read_bytes + hap_bytes < (32u << 20)) {
            memcpy(pin + o_bases, in->bases, read_bytes); memcpy(pin + o_qual, in->qual, read_bytes);
            memcpy(pin + o_ins, in->ins, read_bytes);     memcpy(pin + o_del, in->del, read_bytes);
            memcpy(pin + o_gcp, in->gcp, read_bytes);     memcpy(pin + o_hap, in->hap_bases, hap_bytes);
        } else {
            // a large one-shot batch: the six arrays are staged by six threads (one core copies ~10 GB/s)
            const void* src[6] = {in->bases, in->qual, in->ins, in->del, in->gcp, in->hap_bases};
            uint8_t* dst[6] = {pin + o_bases, pin + o_qual, pin + o_ins, pin + o_del, pin + o_gcp, pin + o_hap};
            const size_t len[6] = {(size_t)read_bytes, (size_t)read_bytes, (size_t)read_bytes, (size_t)read_bytes, (size_t)read_bytes, (size_t)hap_bytes};
            std::thread th[6];
            for (int k = 0; k < 6; ++k) th[k] = std::thread([=] { memcpy(dst[k], src[k], len[k]); });
            for (auto& t : th) t.join();
        }
        HIP_TRY(hipMemcpyAsync(dv, pin, b->in_bytes, hipMemcpyHostToDevice, s));
    } else {
        HIP_TRY(hipMemcpyAsync(b->d_jobs, jobs, n * sizeof(Job), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(b->d_bases, in->bases, read_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(b->d_qual, in->qual, read_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(b->d_ins, in->ins, read_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(b->d_del, in->del, read_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(b->d_gcp, in->gcp, read_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(b->d_hap, in->hap_bases, hap_bytes, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));          // the caller's buffers may go away after we return
        b->host_jobs.clear(); b->host_jobs.shrink_to_fit();
    }
    HIP_TRY(hipEventCreateWithFlags(&b->uploaded, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(b->uploaded, s));       // batch_run makes the compute stream wait on this
    return 0;
}

}  // namespace

int mgx_pairhmm_batch_create(mgx_pairhmm_t* c, const mgx_pairhmm_input_t* in,
                             mgx_pairhmm_batch_t** out) {
    if (!c || !out) { set_error("ctx/out is NULL"); return -EINVAL; }
    *out = nullptr;
    int rc = validate(in);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    BatchPtr b = new_batch();
    if (!b) return -ENOMEM;
    if (!in->pair_read && !in->pair_hap && in->n_reads && in->n_haps) rc = create_cross(c, in, b.get());   // cross-product form
    else rc = create_pairs(c, in, nullptr, b.get());
    if (rc) return rc;
    *out = b.release();
    return 0;
}

int mgx_pairhmm_batch_run(mgx_pairhmm_t* c, mgx_pairhmm_batch_t* b) {
    if (!c || !b) { set_error("ctx/batch is NULL"); return -EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    if (b->n_pairs == 0) { b->ran = true; return 0; }
    hipStream_t s = c->compute;
    const bool timing = (c->flags & MGX_PAIRHMM_TIMING) != 0 && !b->bins.empty();
    const bool force_f64 = (c->flags & MGX_PAIRHMM_FORCE_DOUBLE) != 0;
    hipEvent_t* ev = nullptr;                  // this run's event set
    if (timing) {
        if (b->ev.empty()) {                   // whichever entry point made the batch: allocated on first use
            const size_t per = b->bins.size() * 4;
            b->ev_sets = (uint32_t)std::max<size_t>(1, std::min<size_t>(kMaxEventSets, 1024 / per));
            b->ev.resize(per * b->ev_sets);
            for (auto& e : b->ev) HIP_TRY(hipEventCreate(&e));
        }
        ev = b->ev.data() + (size_t)(b->runs_timed % b->ev_sets) * b->bins.size() * 4;
    }
    if (b->uploaded) HIP_TRY(hipStreamWaitEvent(s, b->uploaded, 0));
    HIP_TRY(hipMemsetAsync(b->d_rerun_count, 0, 64 * sizeof(uint32_t), s));
    HIP_TRY(hipMemsetAsync(b->d_used, 0, b->n_pairs, s));
    // largest class first, classes dealt round-robin to the compute stream and the side streams
    std::vector<size_t> order(b->bins.size());
    for (size_t k = 0; k < order.size(); ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return b->bins[x].cells > b->bins[y].cells; });
    const int n_str = (int)std::min<size_t>((size_t)c->n_streams, b->bins.size());
    if (n_str > 1) {
        HIP_TRY(hipEventRecord(c->ev_fork, s));
        for (int q = 0; q + 1 < n_str; ++q) HIP_TRY(hipStreamWaitEvent(c->aux[q], c->ev_fork, 0));
    }
    // Re-runs in double precision: the "narrow" classes (at most 16 lanes x 8 rows: every read up to 128 bases) append
    // to ONE list (their jobs are a prefix of the job array) that ONE fp64 launch of the 16 x RPLd shape processes --
    // a test case's value does not depend on the shape that computes it; wider classes keep a list and a launch each.
    auto narrow = [](const Bin& bn) { return bn.G <= 16 && bn.G * bn.RPL <= 16 * kNarrowMaxRPL; };
    uint32_t n_narrow_jobs = 0, narrow_max_h = 0, narrow_rows = 0;
    size_t first_narrow = b->bins.size();
    for (size_t k = 0; k < b->bins.size(); ++k) {
        const Bin& bn = b->bins[k];
        if (!narrow(bn)) continue;
        if (first_narrow == b->bins.size()) first_narrow = k;
        n_narrow_jobs = std::max(n_narrow_jobs, bn.job_begin + bn.job_count);
        narrow_max_h = std::max(narrow_max_h, bn.max_h);
        narrow_rows = std::max(narrow_rows, (uint32_t)(bn.G * bn.RPL));
    }
    constexpr int kSharedCount = 63;                       // counter slot of the shared list
    constexpr int kExactCount = 62;                        // ... and of the exact tier's list (second half of d_rerun_list)
    uint32_t* const d_exact_list = b->d_rerun_list + b->n_pairs;
    KernelArgs base{};
    base.jobs = b->d_jobs;
    base.bases = b->d_bases; base.qual = b->d_qual; base.ins = b->d_ins; base.del = b->d_del;
    base.gcp = b->d_gcp; base.hap_bases = b->d_hap;
    base.out_log10 = b->d_out; base.used_f64 = b->d_used;
    base.log10_initial_f = c->log10_initial_f;
    base.log10_initial_d = c->log10_initial_d;
    auto launch_f64 = [&](KernelArgs a, int Gd, int RPLd, uint32_t grid, uint32_t block, uint32_t lds, hipStream_t sk) -> int {
        a.ph2pr = c->d_ph2pr_d; a.mm = c->d_mm_d; a.ph2pr_div3 = c->d_div3_d; a.gap_ratio = c->d_ratio_d;
        a.rerun_list = d_exact_list; a.rerun_count = b->d_rerun_count + kExactCount;      // job_list / n_dyn are set: the fp64 tier appends here
        KernelFn f = a.strip_scratch ? (KernelFn)pairhmm_fwd_strip<double> : pick_kernel<double>(Gd, RPLd);
        if (!f) { set_error("no fp64 kernel for G=%d RPL=%d", Gd, RPLd); return -ENOSYS; }
        hipLaunchKernelGGL(f, dim3(grid), dim3(block), lds, sk, a);
        return 0;
    };
    // Several narrow classes: one multi-class launch per lane-width set instead of one launch per class
    // (MGX_PAIRHMM_MULTI=0 keeps the per-class launches).
    // Only SMALL classes share a launch: measured on an MI355X (profiles/r02_pairhmm_multi_ab.txt) the common launch
    // costs every class the register budget of the hungriest one (131-149 VGPRs: 3 wavefronts per SIMD; 4 when forced,
    // with spills), which outweighs the saved drains for classes that fill the device on their own (ragged 1 M test
    // cases, nine classes of ~29 000 workgroups: 4462 -> 4343 GCUPS) but not for classes of a few thousand workgroups
    // (reads of 20-32 bases: 2764 -> 3009; region-sized batches).  MGX_PAIRHMM_MULTI: 0 never, 1 small classes (default),
    // 2 the same with the build forced to 4 wavefronts per SIMD, 3 every narrow class.
    static const int multi_mode = [] { const char* e = getenv("MGX_PAIRHMM_MULTI"); return e ? atoi(e) : 1; }();
    const uint32_t multi_below = (multi_mode == 3 || multi_mode == 4) ? 0xFFFFFFFFu : (uint32_t)c->n_cu * 64u;     // workgroups; ~5 device fills
    const bool multi_ok = multi_mode != 0;
    std::vector<char> in_multi(b->bins.size(), 0);
    b->acct_cells.assign(b->bins.size(), 0); b->acct_bytes.assign(b->bins.size(), 0); b->acct_multi.assign(b->bins.size(), 0);
    for (size_t k = 0; k < b->bins.size(); ++k) { b->acct_cells[k] = b->bins[k].cells; b->acct_bytes[k] = b->bins[k].alg_bytes; }
    if (multi_ok && !force_f64) {
        for (int gset = 0; gset < 2; ++gset) {
            std::vector<size_t> set;
            for (size_t k = 0; k < b->bins.size(); ++k) {
                const Bin& bn = b->bins[k];
                if (narrow(bn) && bn.RPL <= kNarrowMaxRPL && bn.block == 64 && bn.grid_f32 < multi_below && (gset == 0 ? bn.G == 16 : bn.G < 16)) set.push_back(k);
            }
            if (set.size() < 2 || set.size() > (size_t)kMultiBins) continue;
            // most rows per lane first (the costliest workgroups), wider groups first among equals
            std::stable_sort(set.begin(), set.end(), [&](size_t x, size_t y) {
                const Bin &bx = b->bins[x], &by = b->bins[y];
                return bx.RPL != by.RPL ? bx.RPL > by.RPL : bx.G > by.G;
            });
            MultiArgs m{};
            m.a = base;
            m.a.rerun_list = b->d_rerun_list; m.a.rerun_count = b->d_rerun_count + kSharedCount;
            m.a.job_list = nullptr; m.a.n_dyn = nullptr;
            m.a.ph2pr = c->d_ph2pr_f; m.a.mm = c->d_mm_f; m.a.ph2pr_div3 = c->d_div3_f; m.a.gap_ratio = c->d_ratio_f;
            m.n_bins = (uint32_t)set.size();
            uint32_t blocks = 0, lds = 0;
            size_t book = set[0];
            uint64_t cells = 0, bytes = 0;
            for (size_t q = 0; q < set.size(); ++q) {
                const Bin& bn = b->bins[set[q]];
                m.block_first[q] = blocks; blocks += bn.grid_f32;
                m.job_first[q] = bn.job_begin; m.job_count[q] = bn.job_count; m.lds_stride[q] = bn.lds_stride;
                m.G[q] = (uint8_t)bn.G; m.RPL[q] = (uint8_t)bn.RPL;
                { Bin sc = bn; sc.pk = false; lds = std::max(lds, lds_bytes(sc, true)); }      // the multi-class kernel runs the scalar bodies
                if (bn.cells > b->bins[book].cells) book = set[q];
                cells += bn.cells; bytes += bn.alg_bytes;
                in_multi[set[q]] = 1;
                b->acct_cells[set[q]] = 0; b->acct_bytes[set[q]] = 0;
            }
            m.block_first[set.size()] = blocks;
            b->acct_cells[book] = cells; b->acct_bytes[book] = bytes; b->acct_multi[book] = (int8_t)(1 + gset);
            if (timing) for (size_t q : set) if (q != book) { HIP_TRY(hipEventRecord(ev[4 * q + 0], s)); HIP_TRY(hipEventRecord(ev[4 * q + 1], s)); }
            if (timing) HIP_TRY(hipEventRecord(ev[4 * book + 0], s));
            if (multi_mode == 2 || multi_mode == 4) {
                if (gset == 0) hipLaunchKernelGGL(pairhmm_fwd_multi_occ4<0>, dim3(blocks), dim3(64), lds, s, m);
                else           hipLaunchKernelGGL(pairhmm_fwd_multi_occ4<1>, dim3(blocks), dim3(64), lds, s, m);
            } else {
                if (gset == 0) hipLaunchKernelGGL(pairhmm_fwd_multi<0>, dim3(blocks), dim3(64), lds, s, m);
                else           hipLaunchKernelGGL(pairhmm_fwd_multi<1>, dim3(blocks), dim3(64), lds, s, m);
            }
            if (timing) HIP_TRY(hipEventRecord(ev[4 * book + 1], s));
        }
    }
    for (size_t at = 0; at < order.size(); ++at) {
        const size_t k = order[at];
        const Bin& bin = b->bins[k];
        hipStream_t sk = (int)(at % (size_t)n_str) == 0 ? s : c->aux[at % (size_t)n_str - 1];
        const bool shared = narrow(bin) && !force_f64;
        KernelArgs a = base;
        if (bin.strip) {
            a.strip_stride = (bin.max_h + 63u) & ~63u;
            const size_t need = (size_t)bin.grid_f32 * 6u * a.strip_stride * sizeof(double);
            if (need > c->strip_cap) {
                HIP_TRY(hipStreamSynchronize(s));           // an earlier launch may still be using the old array
                (void)hipFree(c->d_strip); c->d_strip = nullptr; c->strip_cap = 0;
                HIP_TRY(hipMalloc(&c->d_strip, need));
                c->strip_cap = need;
            }
            a.strip_scratch = c->d_strip;
        }
        a.rerun_list = shared ? b->d_rerun_list : b->d_rerun_list + bin.job_begin;
        a.rerun_count = b->d_rerun_count + (shared ? kSharedCount : (int)k);
        a.lds_stride = bin.lds_stride;
        a.job_first = bin.job_begin;
        if (!force_f64 && !in_multi[k]) {
            a.job_list = nullptr; a.n_dyn = nullptr; a.n_static = bin.job_count;
            a.ph2pr = c->d_ph2pr_f; a.mm = c->d_mm_f; a.ph2pr_div3 = c->d_div3_f; a.gap_ratio = c->d_ratio_f;
            KernelFn f = bin.strip ? (KernelFn)pairhmm_fwd_strip<float> : bin.pk ? pick_kernel_pk(bin.G, bin.RPL / 2) : pick_kernel<float>(bin.G, bin.RPL);
            if (!f) { set_error("no fp32 kernel for G=%d RPL=%d", bin.G, bin.RPL); return -ENOSYS; }
            if (timing) HIP_TRY(hipEventRecord(ev[4 * k + 0], sk));
            hipLaunchKernelGGL(f, dim3(bin.grid_f32), dim3(bin.block), lds_bytes(bin, true), sk, a);
            if (timing) HIP_TRY(hipEventRecord(ev[4 * k + 1], sk));
        }
        const bool books_shared = shared && k == first_narrow;      // the shared launch is booked on the first narrow class
        if (timing && !books_shared) HIP_TRY(hipEventRecord(ev[4 * k + 2], sk));
        if (!shared) {
            if (force_f64) { a.job_list = nullptr; a.n_dyn = nullptr; a.n_static = bin.job_count; }
            else { a.job_list = a.rerun_list; a.n_dyn = a.rerun_count; a.n_static = 0; }
            const int rc = launch_f64(a, bin.Gd, bin.RPLd, force_f64 ? bin.grid_f64_all : bin.grid_f64, bin.block, lds_bytes(bin, false), sk);
            if (rc) return rc;
        }
        if (timing && !books_shared) HIP_TRY(hipEventRecord(ev[4 * k + 3], sk));
    }
    for (int q = 0; q + 1 < n_str; ++q) {          // the side streams join the compute stream
        HIP_TRY(hipEventRecord(c->ev_join[q], c->aux[q]));
        HIP_TRY(hipStreamWaitEvent(s, c->ev_join[q], 0));
    }
    if (n_narrow_jobs && !force_f64) {
        KernelArgs a = base;
        a.job_list = b->d_rerun_list; a.n_dyn = b->d_rerun_count + kSharedCount; a.n_static = 0; a.job_first = 0;
        a.lds_stride = (narrow_max_h + 2u * 16u + 8u + 15u) & ~15u;
        const int RPLd = (int)((narrow_rows + 15) / 16);
        const uint32_t grid = std::min<uint32_t>((n_narrow_jobs + 3) / 4, (uint32_t)c->n_cu * 8u);
        if (timing) HIP_TRY(hipEventRecord(ev[4 * first_narrow + 2], s));
        const int rc = launch_f64(a, 16, RPLd, grid, 64, 4u * a.lds_stride, s);
        if (rc) return rc;
        if (timing) HIP_TRY(hipEventRecord(ev[4 * first_narrow + 3], s));
    }
    {
        // The exact tier: whatever the fp64 launches left within reach of the flush-to-zero threshold (next to nothing
        // on real data: log10 likelihoods below about -587), one test case per wavefront, the reference's operation order.
        uint32_t max_h = 0;
        for (const Bin& bn : b->bins) max_h = std::max(max_h, bn.max_h);
        KernelArgs a = base;
        a.ph2pr = c->d_ph2pr_d; a.mm = c->d_mm_d; a.ph2pr_div3 = c->d_div3_d; a.gap_ratio = c->d_ratio_d;
        a.job_list = d_exact_list; a.n_dyn = b->d_rerun_count + kExactCount; a.n_static = 0; a.job_first = 0;
        a.rerun_list = nullptr; a.rerun_count = nullptr;
        a.lds_stride = (max_h + 2u * 64u + 8u + 15u) & ~15u;
        a.strip_stride = (max_h + 63u) & ~63u;
        const size_t per_wg = 6u * (size_t)a.strip_stride * sizeof(double);
        const uint32_t grid = (uint32_t)std::max<size_t>(1, std::min<size_t>({(size_t)b->n_pairs, (size_t)c->n_cu * 2u, (size_t)(256u << 20) / per_wg}));
        const size_t need = (size_t)grid * per_wg;
        if (need > c->strip_cap) {
            HIP_TRY(hipStreamSynchronize(s));
            (void)hipFree(c->d_strip); c->d_strip = nullptr; c->strip_cap = 0;
            HIP_TRY(hipMalloc(&c->d_strip, need));
            c->strip_cap = need;
        }
        a.strip_scratch = c->d_strip;
        hipLaunchKernelGGL(pairhmm_fwd_exact, dim3(grid), dim3(64), a.lds_stride, s, a);
    }
    if (b->has_model && b->d_row_off)
        hipLaunchKernelGGL(pairhmm_normalize_filter_rows, dim3((b->n_reads + 3) / 4), dim3(256), 0, s, b->d_out, b->d_read_len,
                           b->d_row_off, b->d_row_nh, b->n_reads, b->log10_rate, b->max_err, b->d_keep);
    else if (b->has_model)
        hipLaunchKernelGGL(pairhmm_normalize_filter, dim3((b->n_reads + 3) / 4), dim3(256), 0, s, b->d_out, b->d_read_len,
                           b->n_reads, b->n_haps, b->log10_rate, b->max_err, b->d_keep);
    HIP_TRY(hipGetLastError());
    if (timing) b->runs_timed++;
    if (!b->done) HIP_TRY(hipEventCreateWithFlags(&b->done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(b->done, s));
    b->ran = true;
    return 0;
}

int mgx_pairhmm_sync(mgx_pairhmm_t* c) {
    if (!c) { set_error("ctx is NULL"); return -EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->compute));
    return 0;
}

int mgx_pairhmm_batch_stats(mgx_pairhmm_t* c, mgx_pairhmm_batch_t* b, mgx_pairhmm_stats_t* out) {
    if (!c || !b || !out) { set_error("NULL argument"); return -EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->compute));
    mgx_pairhmm_stats_t st = b->stats;
    const bool force_f64 = (c->flags & MGX_PAIRHMM_FORCE_DOUBLE) != 0;
    st.n_launches_f32 = force_f64 ? 0 : (uint32_t)b->bins.size();
    st.n_launches_f64 = (uint32_t)b->bins.size();
    st.n_rerun_f64 = 0;
    if (b->ran && b->n_pairs) {
        std::vector<uint32_t> cnt(64);
        HIP_TRY(hipMemcpy(cnt.data(), b->d_rerun_count, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < b->bins.size(); ++k) st.n_rerun_f64 += force_f64 ? b->bins[k].job_count : cnt[k];
        if (!force_f64) st.n_rerun_f64 += cnt[63];          // the narrow classes' shared list
        st.n_exact = cnt[62];
    }
    st.ms_f32 = st.ms_f64 = st.ms_f32_dominant = 0;
    st.dominant_cells = st.dominant_alg_bytes = 0;
    st.dominant_kernel[0] = 0;
    st.n_runs_timed = 0;
    if ((c->flags & MGX_PAIRHMM_TIMING) && b->ran && !b->ev.empty() && b->runs_timed) {
        // mean over the runs recorded since the previous call (the ring keeps the last ev_sets of them)
        const uint32_t n_sets = std::min(b->runs_timed, b->ev_sets);
        const bool acct = b->acct_cells.size() == b->bins.size();
        auto cells_of = [&](size_t k) { return acct ? b->acct_cells[k] : b->bins[k].cells; };
        size_t dom = 0;
        for (size_t k = 0; k < b->bins.size(); ++k) if (cells_of(k) >= cells_of(dom)) dom = k;
        double f32 = 0, f64 = 0, domms = 0;
        for (uint32_t q = 0; q < n_sets; ++q) {
            const uint32_t set = (b->runs_timed - 1 - q) % b->ev_sets;
            hipEvent_t* ev = b->ev.data() + (size_t)set * b->bins.size() * 4;
            for (size_t k = 0; k < b->bins.size(); ++k) {
                float ms = 0;
                if (!force_f64) {
                    HIP_TRY(hipEventElapsedTime(&ms, ev[4 * k + 0], ev[4 * k + 1]));
                    f32 += ms;
                    if (k == dom) domms += ms;
                }
                float ms2 = 0;
                HIP_TRY(hipEventElapsedTime(&ms2, ev[4 * k + 2], ev[4 * k + 3]));
                f64 += ms2;
            }
        }
        st.n_runs_timed = n_sets;
        st.ms_f32 = (float)(f32 / n_sets); st.ms_f64 = (float)(f64 / n_sets);
        if (!force_f64) {
            st.ms_f32_dominant = (float)(domms / n_sets);
            st.dominant_cells = cells_of(dom);
            st.dominant_alg_bytes = acct ? b->acct_bytes[dom] : b->bins[dom].alg_bytes;
            if (acct && b->acct_multi[dom]) snprintf(st.dominant_kernel, sizeof st.dominant_kernel, "pairhmm_fwd_multi<%d>", b->acct_multi[dom] - 1);
            else if (b->bins[dom].pk) snprintf(st.dominant_kernel, sizeof st.dominant_kernel, "pairhmm_fwd_pk<%d, %d>", b->bins[dom].G, b->bins[dom].RPL / 2);
            else snprintf(st.dominant_kernel, sizeof st.dominant_kernel, "pairhmm_fwd<float, %d, %d>", b->bins[dom].G, b->bins[dom].RPL);
        }
        b->runs_timed = 0;
    }
    *out = st;
    return 0;
}

namespace {
// Results of a batch that has been run, into its pinned mirror (staged batches): waits for this batch's last
// kernel only.  Downloads run on their own stream, so the results of batch k do not queue behind the kernels of
// batch k+1 that were enqueued in the meantime.
int fetch_results(mgx_pairhmm_t* c, mgx_pairhmm_batch_t* b, size_t bytes) {
    hipStream_t s = c->d2h;
    if (b->done) HIP_TRY(hipStreamWaitEvent(s, b->done, 0));
    HIP_TRY(hipMemcpyAsync(b->slab.pin + b->o_out, b->d_out, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}
}  // namespace

int mgx_pairhmm_batch_results(mgx_pairhmm_t* c, mgx_pairhmm_batch_t* b, double* out_log10,
                              uint8_t* used_f64) {
    if (!c || !b) { set_error("ctx/batch is NULL"); return -EINVAL; }
    if (!b->ran) { set_error("batch has not been run"); return -EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    if (b->n_pairs == 0) return 0;
    if (!out_log10) { set_error("out_log10 is NULL"); return -EINVAL; }
    if (b->slab.pin) {
        const int rc = fetch_results(c, b, used_f64 ? b->o_used + b->n_pairs - b->o_out : b->n_pairs * sizeof(double));
        if (rc) return rc;
        memcpy(out_log10, b->slab.pin + b->o_out, b->n_pairs * sizeof(double));
        if (used_f64) memcpy(used_f64, b->slab.pin + b->o_used, b->n_pairs);
        return 0;
    }
    hipStream_t s = c->d2h;
    if (b->done) HIP_TRY(hipStreamWaitEvent(s, b->done, 0));
    HIP_TRY(hipMemcpyAsync(out_log10, b->d_out, b->n_pairs * sizeof(double), hipMemcpyDeviceToHost, s));
    if (used_f64) HIP_TRY(hipMemcpyAsync(used_f64, b->d_used, b->n_pairs, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

void mgx_read_model_defaults(mgx_read_model_t* m) {
    if (!m) return;
    m->pcr_rate_factor = 3;             // CONSERVATIVE, LikelihoodEngineArgumentCollection.h:30
    m->base_quality_threshold = 18;     // PairHMM::BASE_QUALITY_SCORE_THRESHOLD, PairHMM.h:18
    m->constant_gcp = 10;               // gcpHMM
    m->log10_mismapping_rate = -4.5;    // phredScaledGlobalReadMismappingRate = 45
    m->max_error_per_base = 0.02;       // EXPECTED_ERROR_RATE_PER_BASE
}

int mgx_pairhmm_region(mgx_pairhmm_t* c, const mgx_pairhmm_input_t* in, const uint8_t* mapq,
                       const mgx_read_model_t* model, double* out_log10, uint8_t* out_keep) {
    if (!c || !in || !mapq || !model || !out_log10) { set_error("NULL argument"); return -EINVAL; }
    if (in->pair_read || in->pair_hap) { set_error("mgx_pairhmm_region takes the cross-product form (pair arrays NULL)"); return -EINVAL; }
    int rc = validate(in);
    if (rc) return rc;
    if (in->n_reads == 0 || in->n_haps == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    BatchPtr b = new_batch();
    if (!b) return -ENOMEM;
    if ((rc = create_cross(c, in, b.get(), mapq, model))) return rc;
    mgx_pairhmm_batch* raw = b.release();
    rc = mgx_pairhmm_batch_run(c, raw);
    if (!rc) {
        uint8_t* pin = raw->slab.pin;
        rc = [&]() -> int {
            HIP_TRY(hipMemcpyAsync(pin + raw->o_out, raw->d_out, raw->result_bytes, hipMemcpyDeviceToHost, c->compute));
            HIP_TRY(hipStreamSynchronize(c->compute));
            return 0;
        }();
        if (!rc) {
            memcpy(out_log10, pin + raw->o_out, raw->n_pairs * sizeof(double));
            if (out_keep) memcpy(out_keep, pin + raw->o_keep, raw->n_reads);
        }
    }
    mgx_pairhmm_batch_destroy(c, raw);
    return rc;
}

int mgx_pairhmm_compute(mgx_pairhmm_t* c, const mgx_pairhmm_input_t* in, double* out_log10) {
    mgx_pairhmm_batch_t* b = nullptr;
    int rc = mgx_pairhmm_batch_create(c, in, &b);
    if (rc) return rc;
    rc = mgx_pairhmm_batch_run(c, b);
    if (!rc) rc = mgx_pairhmm_batch_results(c, b, out_log10, nullptr);
    mgx_pairhmm_batch_destroy(c, b);
    return rc;
}

int mgx_pairhmm_regions(mgx_pairhmm_t* c, uint32_t n_regions, const mgx_pairhmm_input_t* regions, const uint8_t* const* mapq,
                        const mgx_read_model_t* model, double* const* out_log10, uint8_t* const* out_keep) {
    if (!c || !model || (n_regions && (!regions || !mapq || !out_log10))) { set_error("NULL argument"); return -EINVAL; }
    if (n_regions == 0) return 0;
    int rc;
    for (uint32_t g = 0; g < n_regions; ++g) {
        if ((rc = validate(&regions[g]))) return rc;
        if (regions[g].n_reads * regions[g].n_haps && !out_log10[g]) { set_error("region %u: output pointer is NULL", g); return -EINVAL; }
    }
    HIP_TRY(hipSetDevice(c->device));
    BatchPtr b = new_batch();
    if (!b) return -ENOMEM;
    std::vector<uint64_t> base;
    if ((rc = create_cross_multi(c, n_regions, regions, b.get(), &base, mapq, model))) return rc;
    if (b->n_pairs == 0) return 0;
    mgx_pairhmm_batch* raw = b.release();
    rc = mgx_pairhmm_batch_run(c, raw);
    if (!rc) {
        uint8_t* pin = raw->slab.pin;
        rc = [&]() -> int {
            HIP_TRY(hipMemcpyAsync(pin + raw->o_out, raw->d_out, raw->result_bytes, hipMemcpyDeviceToHost, c->compute));
            HIP_TRY(hipStreamSynchronize(c->compute));
            return 0;
        }();
        if (!rc) {
            const double* all = (const double*)(pin + raw->o_out);
            const uint8_t* keep = pin + raw->o_keep;
            uint64_t r_at = 0;
            for (uint32_t g = 0; g < n_regions; ++g) {
                if (base[g + 1] > base[g]) memcpy(out_log10[g], all + base[g], (base[g + 1] - base[g]) * sizeof(double));
                if (out_keep && out_keep[g] && regions[g].n_reads && regions[g].n_haps) memcpy(out_keep[g], keep + r_at, regions[g].n_reads);
                r_at += regions[g].n_reads;
            }
        }
    }
    mgx_pairhmm_batch_destroy(c, raw);
    return rc;
}

int mgx_pairhmm_compute_regions(mgx_pairhmm_t* c, uint32_t n_regions, const mgx_pairhmm_input_t* regions, double* const* out_log10) {
    if (!c || (n_regions && (!regions || !out_log10))) { set_error("NULL argument"); return -EINVAL; }
    if (n_regions == 0) return 0;
    int rc;
    for (uint32_t g = 0; g < n_regions; ++g) {
        if ((rc = validate(&regions[g]))) return rc;
        if (regions[g].n_reads * regions[g].n_haps && !out_log10[g]) { set_error("region %u: output pointer is NULL", g); return -EINVAL; }
    }
    HIP_TRY(hipSetDevice(c->device));
    // A large call is cut into runs of whole regions of about kChunkPairs (196 608 to 393 216) test cases that go through the context two
    // at a time: while one chunk computes, the next is flattened and uploaded and the previous one's results are
    // scattered -- the queue's pipelining (mgx_pairhmm_queue_run_regions) on the caller's thread alone.
    // (tools/dev_regions_chunk.py, 1000 regions of 40 x 25 on an MI355X: 65 536 test cases per chunk 2 430 GCUPS, 131 072 3 035,
    // 196 608 3 193, 262 144 2 980-3 190, 524 288 2 610, one chunk 2 090: larger chunks mean larger class launches, fewer of them
    // mean less of the flattening hidden behind the kernels)
    // Larger jobs take larger chunks (a sixth of the job, at most 393 216 test cases): 1000 regions of 100 x 50 (5 M test cases)
    // 3 440 GCUPS with 131 072 per chunk, 3 790 with 196 608, 4 160 with 393 216, 3 960 with 524 288; 200 regions of 300 x 100:
    // 3 640 / 4 050 / 4 560 / 4 370; 4000 regions of 40 x 25: 3 460 with 196 608, 3 620 with 393 216 (tools/dev_regions_chunk_shapes.py)
    uint64_t kChunkPairs = 3u << 16;
    {
        uint64_t total = 0;
        for (uint32_t g = 0; g < n_regions; ++g) total += regions[g].n_reads * regions[g].n_haps;
        kChunkPairs = std::min<uint64_t>(6u << 16, std::max<uint64_t>(kChunkPairs, total / 6));
    }
    if (const char* e = getenv("MGX_PAIRHMM_REGION_CHUNK")) { const long long v = atoll(e); if (v > 0) kChunkPairs = (uint64_t)v; }      // A/B
    std::vector<uint32_t> cut(1, 0);
    {
        uint64_t in_chunk = 0;
        for (uint32_t g = 0; g < n_regions; ++g) {
            const uint64_t n = regions[g].n_reads * regions[g].n_haps;
            if (in_chunk && in_chunk + n > kChunkPairs) { cut.push_back(g); in_chunk = 0; }
            in_chunk += n;
        }
        cut.push_back(n_regions);
    }
    struct InFlight { mgx_pairhmm_batch_t* b = nullptr; uint32_t g0 = 0, g1 = 0; std::vector<uint64_t> base; };
    InFlight fl[2];
    auto retire = [&](InFlight& f) -> int {
        if (!f.b) return 0;
        int r = fetch_results(c, f.b, f.b->n_pairs * sizeof(double));
        if (!r) {
            const double* all = (const double*)(f.b->slab.pin + f.b->o_out);
            for (uint32_t g = f.g0; g < f.g1; ++g) {
                const uint64_t a = f.base[g - f.g0], e = f.base[g - f.g0 + 1];
                if (e > a) memcpy(out_log10[g], all + a, (e - a) * sizeof(double));
            }
        }
        mgx_pairhmm_batch_destroy(c, f.b);
        f.b = nullptr;
        return r;
    };
    rc = 0;
    for (size_t k = 0; k + 1 < cut.size() && !rc; ++k) {
        InFlight& f = fl[k & 1];
        rc = retire(f);                                  // the chunk launched two iterations ago
        if (rc) break;
        BatchPtr b = new_batch();
        if (!b) { rc = -ENOMEM; break; }
        f.g0 = cut[k]; f.g1 = cut[k + 1];
        if ((rc = create_cross_multi(c, f.g1 - f.g0, regions + f.g0, b.get(), &f.base))) break;
        if (b->n_pairs == 0) continue;
        f.b = b.release();
        rc = mgx_pairhmm_batch_run(c, f.b);
    }
    for (int q = 0; q < 2; ++q) { const int r = retire(fl[(cut.size() - 1 + q) & 1]); if (!rc) rc = r; }   // oldest first
    return rc;
}

// ---------------------------------------------------------------------------------------------
// Host work queue (BASELINE.json configs[2]).  The reference's worker threads pull the next active
// region off one atomic index (deepmutect/Mutect2Cpp-master/src/main.cpp:254) and share the tail of the
// work at the end (main.cpp:302-315, IntelPairHmm.cc:296-330).  Here the unit pulled is a BATCH of test
// cases: every lane -- a host thread with its own context, i.e. its own streams and recycled pinned
// slabs -- takes the next batch index, plans and packs it (pairhmm_pack.h), uploads, launches and moves on
// to the next one while the device works; a batch's results are fetched when its slot comes round
// again, `depth` batches later.  Several lanes per device keep the PCIe link busy (one core packs
// ~10 GB/s, the link takes ~55); lanes of different devices share the same counter, so a faster GPU
// simply takes more batches.  No collective, no device-to-device traffic.
// ---------------------------------------------------------------------------------------------
struct mgx_pairhmm_queue {
    struct Lane {
        mgx_pairhmm* ctx = nullptr;
        uint32_t dev_slot = 0;
        double pack_s = 0, wait_s = 0;
        uint64_t batches = 0, bytes_h2d = 0, bytes_d2h = 0, cells = 0;
    };
    std::vector<Lane> lanes;
    uint32_t n_devices = 1, depth = 2, batch_pairs = 65536;
    mgx_pairhmm_queue_stats_t stats{};
};

namespace {

struct QueueRun {
    const mgx_pairhmm_input_t* in = nullptr;      // pair-stream mode: one long stream, batches are test-case ranges
    uint64_t lo = 0, hi = 0, n_batches = 0;
    double* out = nullptr;
    uint8_t* used = nullptr;
    // region mode (row F1): many active regions in the cross-product form, batches are runs of whole regions
    const mgx_pairhmm_input_t* regions = nullptr;
    double* const* region_out = nullptr;
    std::vector<uint32_t> chunk;                   // [n_batches + 1] region index boundaries
    std::atomic<uint64_t> next{0};
    std::atomic<int> failed{0};
    std::mutex err_mu;
    int rc = 0;
    std::string err;
    void fail(int code) {
        std::lock_guard<std::mutex> g(err_mu);
        if (!rc) { rc = code; err = mgx_last_error(); }
        failed.store(1);
    }
};

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void queue_lane(mgx_pairhmm_queue* q, mgx_pairhmm_queue::Lane* ln, QueueRun* run) {
    struct Slot { mgx_pairhmm_batch* b = nullptr; uint64_t lo = 0; uint32_t g0 = 0, g1 = 0; std::vector<uint64_t> base; };
    std::vector<Slot> slots(q->depth);
    mgx::PackPlan plan;
    mgx_pairhmm* c = ln->ctx;
    if (hipSetDevice(c->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", c->device); run->fail(-EIO); return; }
    auto retire = [&](Slot& sl) {
        if (!sl.b) return;
        const double t0 = now_s();
        if (!run->failed.load()) {
            int rc = 0;
            if (run->regions) {
                rc = fetch_results(c, sl.b, sl.b->n_pairs * sizeof(double));
                if (!rc) {
                    const double* all = (const double*)(sl.b->slab.pin + sl.b->o_out);
                    for (uint32_t g = sl.g0; g < sl.g1; ++g) {
                        const uint64_t a = sl.base[g - sl.g0], e = sl.base[g - sl.g0 + 1];
                        if (e > a) memcpy(run->region_out[g], all + a, (e - a) * sizeof(double));
                    }
                }
                ln->bytes_d2h += sl.b->n_pairs * 8;
            } else {
                const uint64_t at = sl.lo - run->lo;
                rc = mgx_pairhmm_batch_results(c, sl.b, run->out + at, run->used ? run->used + at : nullptr);
                ln->bytes_d2h += sl.b->n_pairs * (run->used ? 9 : 8);
            }
            if (rc) run->fail(rc);
        }
        mgx_pairhmm_batch_destroy(c, sl.b);
        sl.b = nullptr;
        ln->wait_s += now_s() - t0;
    };
    size_t turn = 0;
    while (!run->failed.load()) {
        const uint64_t k = run->next.fetch_add(1);
        if (k >= run->n_batches) break;
        Slot& sl = slots[turn++ % slots.size()];
        retire(sl);
        if (run->failed.load()) break;
        const double t0 = now_s();
        BatchPtr b = new_batch();
        int rc = b ? 0 : -ENOMEM;
        if (!rc && run->regions) {
            sl.g0 = run->chunk[k]; sl.g1 = run->chunk[k + 1];
            rc = create_cross_multi(c, sl.g1 - sl.g0, run->regions + sl.g0, b.get(), &sl.base);
        } else if (!rc) {
            sl.lo = run->lo + k * q->batch_pairs;
            const uint64_t hi = std::min(run->hi, sl.lo + q->batch_pairs);
            const uint64_t bad = mgx::pack_plan(run->in, sl.lo, hi, &plan);
            if (bad) { set_error("test case %llu: index out of range", (unsigned long long)(sl.lo + bad - 1)); run->fail(-EINVAL); break; }
            rc = create_pairs(c, run->in, &plan, b.get());
        }
        ln->pack_s += now_s() - t0;
        if (!rc && b->n_pairs) rc = mgx_pairhmm_batch_run(c, b.get());
        if (rc) { run->fail(rc); break; }
        if (!b->n_pairs) continue;                  // a chunk of empty regions
        ln->batches++; ln->bytes_h2d += b->in_bytes; ln->cells += b->stats.cells;
        sl.b = b.release();
    }
    for (size_t j = 0; j < slots.size(); ++j) retire(slots[(turn + j) % slots.size()]);    // oldest first
}

int queue_execute(mgx_pairhmm_queue* q, QueueRun& run, uint64_t n_pairs) {
    for (auto& ln : q->lanes) { ln.pack_s = ln.wait_s = 0; ln.batches = ln.bytes_h2d = ln.bytes_d2h = ln.cells = 0; }
    const double t0 = now_s();
    const size_t n_thr = (size_t)std::min<uint64_t>(q->lanes.size(), run.n_batches);
    std::vector<std::thread> th;
    for (size_t i = 1; i < n_thr; ++i) th.emplace_back(queue_lane, q, &q->lanes[i], &run);
    queue_lane(q, &q->lanes[0], &run);               // the caller's thread is lane 0
    for (auto& t : th) t.join();
    q->stats.seconds = now_s() - t0;
    q->stats.n_pairs = n_pairs;
    q->stats.n_batches = run.n_batches;
    for (auto& ln : q->lanes) {
        q->stats.cells += ln.cells; q->stats.bytes_h2d += ln.bytes_h2d; q->stats.bytes_d2h += ln.bytes_d2h;
        q->stats.pack_seconds += ln.pack_s; q->stats.wait_seconds += ln.wait_s;
        q->stats.batches_per_device[ln.dev_slot] += ln.batches;
    }
    if (run.rc) { set_error("%s", run.err.c_str()); return run.rc; }
    return 0;
}

}  // namespace

int mgx_pairhmm_queue_create(const mgx_pairhmm_queue_config_t* cfg, mgx_pairhmm_queue_t** out) {
    if (!out) { set_error("out is NULL"); return -EINVAL; }
    *out = nullptr;
    mgx_pairhmm_queue_config_t d{};
    if (cfg) d = *cfg;
    const int dev0 = 0;
    const uint32_t n_dev = d.n_devices && d.devices ? d.n_devices : 1;
    if (n_dev > 16) { set_error("at most 16 devices per queue"); return -EINVAL; }
    const uint32_t lanes = d.lanes_per_device ? d.lanes_per_device : 4;
    if (lanes > 32) { set_error("at most 32 lanes per device"); return -EINVAL; }
    std::unique_ptr<mgx_pairhmm_queue> q(new (std::nothrow) mgx_pairhmm_queue);
    if (!q) return -ENOMEM;
    q->n_devices = n_dev;
    q->depth = d.depth ? std::min<uint32_t>(d.depth, 8) : 2;
    q->batch_pairs = d.batch_pairs ? d.batch_pairs : 65536;
    for (uint32_t v = 0; v < n_dev; ++v)
        for (uint32_t l = 0; l < lanes; ++l) {
            mgx_pairhmm_queue::Lane ln;
            ln.dev_slot = v;
            const int rc = mgx_pairhmm_create(d.n_devices && d.devices ? d.devices[v] : dev0, d.flags, &ln.ctx);
            if (rc) { for (auto& x : q->lanes) mgx_pairhmm_destroy(x.ctx); return rc; }
            q->lanes.push_back(ln);
        }
    *out = q.release();
    return 0;
}

void mgx_pairhmm_queue_destroy(mgx_pairhmm_queue_t* q) {
    if (!q) return;
    for (auto& ln : q->lanes) mgx_pairhmm_destroy(ln.ctx);
    delete q;
}

int mgx_pairhmm_queue_run_range(mgx_pairhmm_queue_t* q, const mgx_pairhmm_input_t* in, uint64_t pair_begin, uint64_t pair_end,
                                double* out_log10, uint8_t* used_f64) {
    if (!q) { set_error("queue is NULL"); return -EINVAL; }
    int rc = validate(in);
    if (rc) return rc;
    const uint64_t total = mgx::pack_n_pairs(in);
    if (pair_begin > pair_end || pair_end > total) { set_error("test-case range [%llu, %llu) outside the stream of %llu", (unsigned long long)pair_begin, (unsigned long long)pair_end, (unsigned long long)total); return -EINVAL; }
    q->stats = mgx_pairhmm_queue_stats_t{};
    q->stats.n_lanes = (uint32_t)q->lanes.size();
    if (pair_begin == pair_end) return 0;
    if (!out_log10) { set_error("out_log10 is NULL"); return -EINVAL; }
    QueueRun run;
    run.in = in; run.lo = pair_begin; run.hi = pair_end; run.out = out_log10; run.used = used_f64;
    run.n_batches = (pair_end - pair_begin + q->batch_pairs - 1) / q->batch_pairs;
    return queue_execute(q, run, pair_end - pair_begin);
}

int mgx_pairhmm_queue_run_regions(mgx_pairhmm_queue_t* q, uint32_t n_regions, const mgx_pairhmm_input_t* regions, double* const* out_log10) {
    if (!q || (n_regions && (!regions || !out_log10))) { set_error("NULL argument"); return -EINVAL; }
    q->stats = mgx_pairhmm_queue_stats_t{};
    q->stats.n_lanes = (uint32_t)q->lanes.size();
    QueueRun run;
    run.regions = regions; run.region_out = out_log10;
    // batches are runs of whole regions holding about batch_pairs test cases
    uint64_t total = 0, in_chunk = 0;
    run.chunk.push_back(0);
    for (uint32_t g = 0; g < n_regions; ++g) {
        int rc = validate(&regions[g]);
        if (rc) return rc;
        if (regions[g].pair_read || regions[g].pair_hap) { set_error("region %u: regions are given in the cross-product form (pair arrays NULL)", g); return -EINVAL; }
        const uint64_t n = regions[g].n_reads * regions[g].n_haps;
        if (n && !out_log10[g]) { set_error("region %u: output pointer is NULL", g); return -EINVAL; }
        if (in_chunk && in_chunk + n > q->batch_pairs) { run.chunk.push_back(g); in_chunk = 0; }
        in_chunk += n; total += n;
    }
    if (n_regions) run.chunk.push_back(n_regions);
    run.n_batches = run.chunk.size() - 1;
    if (total == 0) return 0;
    return queue_execute(q, run, total);
}

int mgx_pairhmm_queue_run(mgx_pairhmm_queue_t* q, const mgx_pairhmm_input_t* in, double* out_log10, uint8_t* used_f64) {
    if (!in) { set_error("input is NULL"); return -EINVAL; }
    return mgx_pairhmm_queue_run_range(q, in, 0, mgx::pack_n_pairs(in), out_log10, used_f64);
}

int mgx_pairhmm_queue_stats(mgx_pairhmm_queue_t* q, mgx_pairhmm_queue_stats_t* out) {
    if (!q || !out) { set_error("NULL argument"); return -EINVAL; }
    *out = q->stats;
    return 0;
}

}  // extern "C"
