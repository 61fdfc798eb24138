// mgx_pairhmm.hip -- host side of the PairHMM C ABI (include/mgx_pairhmm.h) for MI355X.
//
// Replaces, behind a plain C boundary, the reference's native PairHMM layer
// (deepmutect/Mutect2Cpp-master/src/intel/pairhmm/IntelPairHmm.cc:202-351): table set-up,
// float-first / double-fallback policy and the per-test-case loop.  The per-test-case loop
// becomes: bin test cases by read-length class, sort each bin by haplotype length, launch one
// fp32 kernel per bin, then one fp64 kernel per bin over the device-side re-run list.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/mgx_pairhmm.h"
#include "mgx_common.h"
#include "mgx_tables.h"
#include "pairhmm_kernels.hip.inc"

using mgx::set_error;

namespace {

constexpr int kMaxRowsG16 = 16 * 8;
constexpr int kMaxRowsG64 = 64 * 8;
constexpr uint32_t kMaxLdsPerBlock = 64 * 1024;

struct Bin {
    int G = 0, RPL = 0;
    uint32_t job_begin = 0, job_count = 0;
    uint32_t max_h = 0;
    uint64_t cells = 0, alg_bytes = 0;
    // launch geometry
    uint32_t block = 256, lds_stride = 0, grid_f32 = 0, grid_f64 = 0;
};

}  // namespace

struct mgx_pairhmm {
    int device = 0;
    unsigned flags = 0;
    hipStream_t compute = nullptr, copy = nullptr;
    float* d_ph2pr_f = nullptr; float* d_mm_f = nullptr;
    double* d_ph2pr_d = nullptr; double* d_mm_d = nullptr;
    float log10_initial_f = 0; double log10_initial_d = 0;
    int n_cu = 256;
};

struct mgx_pairhmm_batch {
    uint64_t n_pairs = 0;
    std::vector<Bin> bins;
    // device buffers
    uint8_t *d_bases = nullptr, *d_qual = nullptr, *d_ins = nullptr, *d_del = nullptr,
            *d_gcp = nullptr, *d_hap = nullptr, *d_used = nullptr;
    Job* d_jobs = nullptr;
    uint32_t* d_rerun_list = nullptr;
    uint32_t* d_rerun_count = nullptr;
    double* d_out = nullptr;
    // timing
    std::vector<hipEvent_t> ev;    // 4 per bin: f32 start/stop, f64 start/stop
    bool ran = false;
    mgx_pairhmm_stats_t stats{};
};

namespace {

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

// (G, RPL) class of a read of R rows; G = 0 if unsupported.
inline void shape_of(uint32_t R, int* G, int* RPL) {
    if (R <= (uint32_t)kMaxRowsG16) { *G = 16; *RPL = (int)((R + 15) / 16); }
    else if (R <= (uint32_t)kMaxRowsG64) { *G = 64; *RPL = (int)((R + 63) / 64); }
    else { *G = 0; *RPL = 0; }
}
// dynamic LDS of one block: per-wavefront emission table (fp32 only) + per-group haplotype codes
inline uint32_t lds_bytes(const Bin& bin, bool f32) {
    const uint32_t waves = bin.block / 64u, groups = bin.block / (uint32_t)bin.G;
    const uint32_t etab = f32 ? (uint32_t)((bin.RPL + 1) / 2) * kNumCodes * 512u : 0u;
    return waves * etab + groups * bin.lds_stride;
}
inline int bin_index(int G, int RPL) { return (G == 16 ? 0 : 8) + RPL - 1; }

int validate(const mgx_pairhmm_input_t* in) {
    if (!in) { set_error("input is NULL"); return -EINVAL; }
    if (in->n_pairs == 0) return 0;
    if (!in->read_off || !in->hap_off || !in->bases || !in->qual || !in->ins || !in->del ||
        !in->gcp || !in->hap_bases || !in->pair_read || !in->pair_hap) {
        set_error("a required input array is NULL");
        return -EINVAL;
    }
    if (in->n_pairs > 0xFFFFFFF0ull) { set_error("more than 2^32 test cases in one batch"); return -E2BIG; }
    return 0;
}

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
    if (device < 0 || device >= n_dev) { set_error("device %d out of range (0..%d)", device, n_dev - 1); return -EINVAL; }
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<mgx_pairhmm> c(new (std::nothrow) mgx_pairhmm);
    if (!c) return -ENOMEM;
    c->device = device;
    c->flags = flags;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount;
    HIP_TRY(hipStreamCreateWithFlags(&c->compute, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
    const auto& tf = mgx::tables<float>();
    const auto& td = mgx::tables<double>();
    int rc;
    if ((rc = upload(&c->d_ph2pr_f, tf.ph2pr.data(), tf.ph2pr.size() * 4, c->compute))) return rc;
    if ((rc = upload(&c->d_mm_f, tf.mm.data(), tf.mm.size() * 4, c->compute))) return rc;
    if ((rc = upload(&c->d_ph2pr_d, td.ph2pr.data(), td.ph2pr.size() * 8, c->compute))) return rc;
    if ((rc = upload(&c->d_mm_d, td.mm.data(), td.mm.size() * 8, c->compute))) return rc;
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
    if (c->compute) (void)hipStreamDestroy(c->compute);
    if (c->copy) (void)hipStreamDestroy(c->copy);
    delete c;
}

void mgx_pairhmm_batch_destroy(mgx_pairhmm_t* c, mgx_pairhmm_batch_t* b) {
    if (!b) return;
    if (c) (void)hipSetDevice(c->device);
    (void)hipFree(b->d_bases); (void)hipFree(b->d_qual); (void)hipFree(b->d_ins);
    (void)hipFree(b->d_del); (void)hipFree(b->d_gcp); (void)hipFree(b->d_hap);
    (void)hipFree(b->d_used); (void)hipFree(b->d_jobs); (void)hipFree(b->d_rerun_list);
    (void)hipFree(b->d_rerun_count); (void)hipFree(b->d_out);
    for (auto e : b->ev) (void)hipEventDestroy(e);
    delete b;
}

int mgx_pairhmm_batch_create(mgx_pairhmm_t* c, const mgx_pairhmm_input_t* in,
                             mgx_pairhmm_batch_t** out) {
    if (!c || !out) { set_error("ctx/out is NULL"); return -EINVAL; }
    *out = nullptr;
    int rc = validate(in);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    std::unique_ptr<mgx_pairhmm_batch, void (*)(mgx_pairhmm_batch*)> b(
        new (std::nothrow) mgx_pairhmm_batch, [](mgx_pairhmm_batch* p) { mgx_pairhmm_batch_destroy(nullptr, p); });
    if (!b) return -ENOMEM;
    const uint64_t n = in->n_pairs;
    b->n_pairs = n;
    b->stats.n_pairs = n;

    // ---- bin by (G, RPL), then counting-sort every bin by haplotype length -------------
    constexpr int kBins = 16;
    std::vector<uint32_t> bin_of(n);
    uint64_t count[kBins] = {0};
    uint32_t max_h = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t r = in->pair_read[i], h = in->pair_hap[i];
        if (r >= in->n_reads || h >= in->n_haps) { set_error("test case %llu: index out of range", (unsigned long long)i); return -EINVAL; }
        const uint64_t R = in->read_off[r + 1] - in->read_off[r];
        const uint64_t H = in->hap_off[h + 1] - in->hap_off[h];
        if (R == 0 || H == 0) { set_error("test case %llu: empty read or haplotype", (unsigned long long)i); return -EINVAL; }
        int G, RPL;
        shape_of((uint32_t)std::min<uint64_t>(R, 0xFFFFFFFFull), &G, &RPL);
        if (G == 0) { set_error("test case %llu: read of %llu bases exceeds the %d-row limit", (unsigned long long)i, (unsigned long long)R, kMaxRowsG64); return -E2BIG; }
        if (H > 0x7FFFFFF0ull) { set_error("haplotype too long"); return -E2BIG; }
        const int bi = bin_index(G, RPL);
        bin_of[i] = (uint32_t)bi;
        count[bi]++;
        max_h = std::max<uint32_t>(max_h, (uint32_t)H);
        b->stats.cells += R * H;
        b->stats.alg_bytes += 5 * R + H + 4;
    }
    std::vector<Job> jobs(n);
    {
        // key = (bin, H): counting sort on H inside each bin keeps the wavefront's groups
        // (consecutive jobs) at near-equal step counts.
        std::vector<uint64_t> bin_start(kBins + 1, 0);
        for (int k = 0; k < kBins; ++k) bin_start[k + 1] = bin_start[k] + count[k];
        std::vector<std::vector<uint32_t>> hist(kBins);
        for (int k = 0; k < kBins; ++k) if (count[k]) hist[k].assign((size_t)max_h + 2, 0);
        for (uint64_t i = 0; i < n; ++i) {
            const uint32_t h = in->pair_hap[i];
            const uint32_t H = (uint32_t)(in->hap_off[h + 1] - in->hap_off[h]);
            hist[bin_of[i]][H + 1]++;
        }
        for (int k = 0; k < kBins; ++k)
            for (size_t x = 1; x < hist[k].size(); ++x) hist[k][x] += hist[k][x - 1];
        for (uint64_t i = 0; i < n; ++i) {
            const uint32_t r = in->pair_read[i], h = in->pair_hap[i];
            const uint32_t R = (uint32_t)(in->read_off[r + 1] - in->read_off[r]);
            const uint32_t H = (uint32_t)(in->hap_off[h + 1] - in->hap_off[h]);
            const int k = (int)bin_of[i];
            Job& jb = jobs[bin_start[k] + hist[k][H]++];
            jb.read_off = in->read_off[r]; jb.hap_off = in->hap_off[h];
            jb.R = R; jb.H = H; jb.pair = (uint32_t)i; jb.pad_ = 0;
        }
        for (int k = 0; k < kBins; ++k) {
            if (!count[k]) continue;
            Bin bin;
            bin.G = k < 8 ? 16 : 64;
            bin.RPL = (k % 8) + 1;
            bin.job_begin = (uint32_t)bin_start[k];
            bin.job_count = (uint32_t)count[k];
            for (uint64_t q = bin_start[k]; q < bin_start[k + 1]; ++q) {
                bin.max_h = std::max(bin.max_h, jobs[q].H);
                bin.cells += (uint64_t)jobs[q].R * jobs[q].H;
                bin.alg_bytes += 5ull * jobs[q].R + jobs[q].H + 4;
            }
            // LDS per group: G pad + codes + G pad + prefetch slack (see the kernel's staging loop)
            bin.lds_stride = (bin.max_h + 2u * (uint32_t)bin.G + 8u + 15u) & ~15u;
            bin.block = 128;                  // 2 wavefronts: fine-grained LDS/VGPR packing per CU
            while (bin.block > 64u && lds_bytes(bin, true) > kMaxLdsPerBlock) bin.block /= 2;
            if (lds_bytes(bin, true) > 160u * 1024u) {
                set_error("haplotype of %u bases does not fit the LDS staging buffer", bin.max_h);
                return -E2BIG;
            }
            const uint32_t gpb = bin.block / bin.G;
            bin.grid_f32 = (bin.job_count + gpb - 1) / gpb;
            bin.grid_f64 = std::min<uint32_t>(bin.grid_f32, (uint32_t)c->n_cu * 8u);
            b->bins.push_back(bin);
        }
    }

    // ---- upload ---------------------------------------------------------------------
    const uint64_t read_bytes = in->read_off[in->n_reads];
    const uint64_t hap_bytes = in->hap_off[in->n_haps];
    hipStream_t s = c->copy;
    if ((rc = upload(&b->d_bases, in->bases, read_bytes, s))) return rc;
    if ((rc = upload(&b->d_qual, in->qual, read_bytes, s))) return rc;
    if ((rc = upload(&b->d_ins, in->ins, read_bytes, s))) return rc;
    if ((rc = upload(&b->d_del, in->del, read_bytes, s))) return rc;
    if ((rc = upload(&b->d_gcp, in->gcp, read_bytes, s))) return rc;
    if ((rc = upload(&b->d_hap, in->hap_bases, hap_bytes, s))) return rc;
    if ((rc = upload(&b->d_jobs, jobs.data(), n * sizeof(Job), s))) return rc;
    if ((rc = upload(&b->d_rerun_list, nullptr, n * sizeof(uint32_t), s))) return rc;
    if ((rc = upload(&b->d_rerun_count, nullptr, 64 * sizeof(uint32_t), s))) return rc;
    if ((rc = upload(&b->d_out, nullptr, n * sizeof(double), s))) return rc;
    if ((rc = upload(&b->d_used, nullptr, n, s))) return rc;
    HIP_TRY(hipStreamSynchronize(s));   // jobs vector goes out of scope; batch is now resident
    if (c->flags & MGX_PAIRHMM_TIMING) {
        b->ev.resize(b->bins.size() * 4);
        for (auto& e : b->ev) HIP_TRY(hipEventCreate(&e));
    }
    *out = b.release();
    return 0;
}

int mgx_pairhmm_batch_run(mgx_pairhmm_t* c, mgx_pairhmm_batch_t* b) {
    if (!c || !b) { set_error("ctx/batch is NULL"); return -EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    if (b->n_pairs == 0) { b->ran = true; return 0; }
    hipStream_t s = c->compute;
    const bool timing = (c->flags & MGX_PAIRHMM_TIMING) != 0;
    const bool force_f64 = (c->flags & MGX_PAIRHMM_FORCE_DOUBLE) != 0;
    HIP_TRY(hipMemsetAsync(b->d_rerun_count, 0, 64 * sizeof(uint32_t), s));
    HIP_TRY(hipMemsetAsync(b->d_used, 0, b->n_pairs, s));
    for (size_t k = 0; k < b->bins.size(); ++k) {
        const Bin& bin = b->bins[k];
        KernelArgs a{};
        a.jobs = b->d_jobs + bin.job_begin;
        a.bases = b->d_bases; a.qual = b->d_qual; a.ins = b->d_ins; a.del = b->d_del;
        a.gcp = b->d_gcp; a.hap_bases = b->d_hap;
        a.out_log10 = b->d_out; a.used_f64 = b->d_used;
        a.rerun_list = b->d_rerun_list + bin.job_begin;
        a.rerun_count = b->d_rerun_count + k;
        a.lds_stride = bin.lds_stride;
        a.log10_initial_f = c->log10_initial_f;
        a.log10_initial_d = c->log10_initial_d;
        if (!force_f64) {
            a.job_list = nullptr; a.n_dyn = nullptr; a.n_static = bin.job_count;
            a.ph2pr = c->d_ph2pr_f; a.mm = c->d_mm_f;
            KernelFn f = pick_kernel<float>(bin.G, bin.RPL);
            if (!f) { set_error("no fp32 kernel for G=%d RPL=%d", bin.G, bin.RPL); return -ENOSYS; }
            if (timing) HIP_TRY(hipEventRecord(b->ev[4 * k + 0], s));
            hipLaunchKernelGGL(f, dim3(bin.grid_f32), dim3(bin.block), lds_bytes(bin, true), s, a);
            if (timing) HIP_TRY(hipEventRecord(b->ev[4 * k + 1], s));
        }
        {
            a.ph2pr = c->d_ph2pr_d; a.mm = c->d_mm_d;
            if (force_f64) { a.job_list = nullptr; a.n_dyn = nullptr; a.n_static = bin.job_count; }
            else { a.job_list = b->d_rerun_list + bin.job_begin; a.n_dyn = b->d_rerun_count + k; a.n_static = 0; }
            KernelFn f = pick_kernel<double>(bin.G, bin.RPL);
            if (!f) { set_error("no fp64 kernel for G=%d RPL=%d", bin.G, bin.RPL); return -ENOSYS; }
            if (timing) HIP_TRY(hipEventRecord(b->ev[4 * k + 2], s));
            hipLaunchKernelGGL(f, dim3(force_f64 ? bin.grid_f32 : bin.grid_f64), dim3(bin.block), lds_bytes(bin, false), s, a);
            if (timing) HIP_TRY(hipEventRecord(b->ev[4 * k + 3], s));
        }
    }
    HIP_TRY(hipGetLastError());
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
    }
    st.ms_f32 = st.ms_f64 = st.ms_f32_dominant = 0;
    st.dominant_cells = st.dominant_alg_bytes = 0;
    st.dominant_kernel[0] = 0;
    if ((c->flags & MGX_PAIRHMM_TIMING) && b->ran && !b->ev.empty()) {
        uint64_t best_cells = 0;
        for (size_t k = 0; k < b->bins.size(); ++k) {
            float ms = 0;
            if (!force_f64) {
                HIP_TRY(hipEventElapsedTime(&ms, b->ev[4 * k + 0], b->ev[4 * k + 1]));
                st.ms_f32 += ms;
                if (b->bins[k].cells >= best_cells) {
                    best_cells = b->bins[k].cells;
                    st.ms_f32_dominant = ms;
                    st.dominant_cells = b->bins[k].cells;
                    st.dominant_alg_bytes = b->bins[k].alg_bytes;
                    snprintf(st.dominant_kernel, sizeof st.dominant_kernel,
                             "pairhmm_fwd<float, %d, %d>", b->bins[k].G, b->bins[k].RPL);
                }
            }
            float ms2 = 0;
            HIP_TRY(hipEventElapsedTime(&ms2, b->ev[4 * k + 2], b->ev[4 * k + 3]));
            st.ms_f64 += ms2;
        }
    }
    *out = st;
    return 0;
}

int mgx_pairhmm_batch_results(mgx_pairhmm_t* c, mgx_pairhmm_batch_t* b, double* out_log10,
                              uint8_t* used_f64) {
    if (!c || !b) { set_error("ctx/batch is NULL"); return -EINVAL; }
    if (!b->ran) { set_error("batch has not been run"); return -EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    if (b->n_pairs == 0) return 0;
    if (!out_log10) { set_error("out_log10 is NULL"); return -EINVAL; }
    HIP_TRY(hipMemcpyAsync(out_log10, b->d_out, b->n_pairs * sizeof(double), hipMemcpyDeviceToHost, c->compute));
    if (used_f64) HIP_TRY(hipMemcpyAsync(used_f64, b->d_used, b->n_pairs, hipMemcpyDeviceToHost, c->compute));
    HIP_TRY(hipStreamSynchronize(c->compute));
    return 0;
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

}  // extern "C"
