/*
 * mgx_pairhmm.h -- C ABI of the MI355X PairHMM likelihood engine (libmgx.so).
 *
 * Drop-in seam: this is what the reference's ex-JNI "native" layer would bind instead of the
 * Intel GKL AVX kernels.  Reference interface replaced (paths relative to
 * deepmutect/Mutect2Cpp-master/src/):
 *
 *   mgx_pairhmm_create            <- initNative(bool use_double, int max_threads)
 *                                    intel/pairhmm/IntelPairHmm.h:37, IntelPairHmm.cc:202-256
 *   mgx_pairhmm_compute           <- computeLikelihoodsNative(vector<testcase>&, vector<double>&)
 *                                    intel/pairhmm/IntelPairHmm.h:39, IntelPairHmm.cc:259-293
 *                                    and computeLikelihoodsNative_concurrent_i  (:332-351), the
 *                                    per-test-case loop of VectorLoglessPairHMM.cpp:118-119
 *   mgx_pairhmm_compute_regions   <- that call for several active regions at once (row F1)
 *   mgx_pairhmm_batch_*           <- the same call split into upload / run / download so a caller
 *                                    can keep several active regions in flight (replaces the
 *                                    tail-phase work sharing, IntelPairHmm.cc:296-330, 659-693)
 *   mgx_pairhmm_destroy           <- doneNative (IntelPairHmm.cc:190-198, a no-op there)
 *
 * A "test case" (pairhmm_common.h:45-57: haplen, hap, ReadForPairHMM) is flattened to indices
 * into packed read / haplotype arrays so that nothing but plain pointers and sizes crosses the
 * boundary.  Per-read inputs are the five byte arrays ReadForPairHMM's constructor takes
 * (haplotypecaller/ReadForPairHMM.cpp:18-38): bases, base quals, insertion GOP, deletion GOP,
 * gap-continuation penalty.  Every quality byte is masked with 127 on the device exactly as the
 * reference does; bases are ASCII, any byte other than A/C/G/T/N is treated as 'A'
 * (pairhmm_common.h:75-81).  Reads may be up to 2^20 bases long (beyond 1024 a strip-mined kernel is used),
 * haplotypes up to about 40 000 bases;
 * longer sequences are rejected with -E2BIG.
 *
 * All functions return 0 on success or a negative errno-style code; no exception crosses the
 * boundary.  mgx_last_error() returns a thread-local message for the last failure.
 * Buffers passed in are caller-owned and only read during the call; the library never frees
 * them.  A context is bound to one device and one compute stream and may be used by one host
 * thread at a time; use one context per worker thread (the reference also keeps one
 * VectorLoglessPairHMM per worker thread, Mutect2Engine.cpp:27-29).
 */
#ifndef MGX_PAIRHMM_H
#define MGX_PAIRHMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mgx_pairhmm mgx_pairhmm_t;
typedef struct mgx_pairhmm_batch mgx_pairhmm_batch_t;

/* flags for mgx_pairhmm_create */
#define MGX_PAIRHMM_FORCE_DOUBLE 1u /* PairHMMNativeArgumentCollection.useDoublePrecision */
#define MGX_PAIRHMM_TIMING       2u /* record HIP events around every kernel launch */
#define MGX_PAIRHMM_PACKED_FP32  4u /* classes with an even number of rows per lane run pairhmm_fwd_pk (two rows per v_pk_*_f32
                                     * instruction).  Bit-identical results; measured SLOWER than the default kernel on every
                                     * shape (a packed instruction occupies the fp32 pipe as long as two scalar
                                     * ones), so it is off unless asked for (or MGX_PAIRHMM_PK=1) */
/* bits 8..15 (both mgx_pairhmm_create and mgx_sortdedup_create): optional CU partition, an 8-bit
 * pattern repeated over the CU index; 0 or 0xFF = all CUs.  MGX_CU_PATTERN(0x3F) keeps 6 CUs of 8. */
#define MGX_CU_PATTERN(p) (((unsigned)(p) & 0xFFu) << 8)
/* bit 16 (both create calls): the context's streams get the highest priority the device offers -- for the HBM-bound sort
 * pipeline when it shares a GPU with PairHMM batches (BASELINE.json configs[4]): its short kernels then take the compute units
 * PairHMM workgroups free every ~20 us instead of queueing behind them.  Ignored together with a CU pattern. */
#define MGX_STREAM_HIGH_PRIORITY (1u << 16)

/* Packed host-side description of one batch of test cases. */
typedef struct mgx_pairhmm_input {
    uint64_t n_reads;
    const uint64_t* read_off;  /* [n_reads + 1] prefix offsets into the five read arrays */
    const uint8_t* bases;      /* ASCII read bases */
    const uint8_t* qual;       /* base qualities */
    const uint8_t* ins;        /* insertion gap-open penalties */
    const uint8_t* del;        /* deletion gap-open penalties */
    const uint8_t* gcp;        /* gap continuation penalties */
    uint64_t n_haps;
    const uint64_t* hap_off;   /* [n_haps + 1] prefix offsets into hap_bases */
    const uint8_t* hap_bases;  /* ASCII haplotype bases */
    uint64_t n_pairs;
    const uint32_t* pair_read; /* [n_pairs] read index of test case i */
    const uint32_t* pair_hap;  /* [n_pairs] haplotype index of test case i */
    /* Cross-product form: pair_read == pair_hap == NULL means "every read against every
     * haplotype", n_reads * n_haps test cases with out_log10[r * n_haps + h] -- the list
     * VectorLoglessPairHMM builds for a region (VectorLoglessPairHMM.cpp:88-93); n_pairs is then
     * ignored, host work is O(n_reads + n_haps) and the job list is generated on the device. */
} mgx_pairhmm_input_t;

typedef struct mgx_pairhmm_stats {
    uint64_t n_pairs;
    uint64_t cells;            /* sum of R*H over the batch */
    uint64_t alg_bytes;        /* sum of 5R + H + 4 */
    uint64_t n_rerun_f64;      /* test cases whose fp32 result was < 1e-28f (last run) */
    uint32_t n_launches_f32;   /* kernel launches of the fp32 recurrence per run */
    uint32_t n_launches_f64;
    /* valid only with MGX_PAIRHMM_TIMING: HIP events recorded on the compute stream around every kernel
     * of EVERY mgx_pairhmm_batch_run; the figures are means per run over the runs since the previous
     * mgx_pairhmm_batch_stats call (the last 64 at most), n_runs_timed says how many */
    uint32_t n_runs_timed;
    float ms_f32;              /* sum of the fp32 recurrence kernels' durations */
    float ms_f64;              /* sum of the fp64 re-run kernels' durations */
    float ms_f32_dominant;     /* duration of the largest fp32 launch ... */
    uint64_t dominant_cells;   /* ... and the cells / algorithmic bytes it processed */
    uint64_t dominant_alg_bytes;
    char dominant_kernel[64];  /* its name as rocprofv3 prints it (prefix) */
    uint64_t n_exact;          /* test cases whose fp64 result was < 1e-280 (in reach of the flush-to-zero threshold) and
                                * that were computed a third time in the reference's exact operation order (last run) */
} mgx_pairhmm_stats_t;

const char* mgx_last_error(void);

#define MGX_DEVICE_AUTO (-1)   /* device argument of the create functions: next GPU, round-robin per process */

/* device: HIP device ordinal or MGX_DEVICE_AUTO (one context per worker thread then spreads the
 * threads over the GPUs of the node -- the host work queue of the reference's threadFunc,
 * main.cpp:254, needs nothing else).  Builds the Context<float>/Context<double> tables
 * (intel/pairhmm/Context.h) on the host and uploads them. */
int mgx_pairhmm_create(int device, unsigned flags, mgx_pairhmm_t** out);
void mgx_pairhmm_destroy(mgx_pairhmm_t* ctx);

/* One shot: host buffers in, log10 likelihoods out (out_log10[i] for test case i). */
int mgx_pairhmm_compute(mgx_pairhmm_t* ctx, const mgx_pairhmm_input_t* in, double* out_log10);

/* several active regions in ONE device batch (one upload, one set of launches,
 * one download) -- what replaces the reference's tail-phase work sharing between worker threads
 * (main.cpp:302-315, IntelPairHmm.cc:296-330).  Every region is given in the cross-product form
 * (pair arrays NULL); out_log10[g] receives region g's [n_reads][n_haps] block.  Values are identical
 * to one mgx_pairhmm_compute call per region (a test case's result does not depend on its batch). */
int mgx_pairhmm_compute_regions(mgx_pairhmm_t* ctx, uint32_t n_regions, const mgx_pairhmm_input_t* regions,
                                double* const* out_log10);

/* Staged form.  batch_create bins the test cases by shape, stages the packed arrays through
 * pinned memory and uploads them on the context's copy stream; after it returns the batch is
 * resident in HBM.  batch_run only enqueues kernels on the compute stream (asynchronous).
 * batch_results waits for the stream and copies the results back (used_f64 may be NULL). */
int mgx_pairhmm_batch_create(mgx_pairhmm_t* ctx, const mgx_pairhmm_input_t* in,
                             mgx_pairhmm_batch_t** out);
int mgx_pairhmm_batch_run(mgx_pairhmm_t* ctx, mgx_pairhmm_batch_t* batch);
int mgx_pairhmm_batch_results(mgx_pairhmm_t* ctx, mgx_pairhmm_batch_t* batch, double* out_log10,
                              uint8_t* used_f64);
int mgx_pairhmm_batch_stats(mgx_pairhmm_t* ctx, mgx_pairhmm_batch_t* batch,
                            mgx_pairhmm_stats_t* out);
void mgx_pairhmm_batch_destroy(mgx_pairhmm_t* ctx, mgx_pairhmm_batch_t* batch);
int mgx_pairhmm_sync(mgx_pairhmm_t* ctx);

/* Whole-region form (row F2 of SURVEY.md 8f): what PairHMMLikelihoodCalculationEngine::
 * computeReadLikelihoods does for one sample (haplotypecaller/PairHMMLikelihoodCalculationEngine.cpp:63-97)
 * in one call -- on the device, between one upload and one download:
 *   modifyReadQualities   PCR indel error model from tandem-repeat length, base quality capped by
 *                         MAPQ, low qualities squashed to 6                      (:123-282)
 *   gap continuation      constant penalty                                     (:284-292)
 *   PairHMM               every read x every haplotype (`in` in cross-product form, raw qualities)
 *   normalizeLikelihoods  cap every haplotype at best + log10 mismapping rate  (AlleleLikelihoods.h:372-391)
 *   filterPoorlyModeledEvidence   out_keep[r] = 0 for reads it would drop      (AlleleLikelihoods.h:404-419)
 * out_log10 is [n_reads][n_haps]; mapq is one byte per read.  Parity of this row is pinned by the
 * oracle's restatement only (the reference TUs need the SAMRecord / VariantContext model). */
typedef struct mgx_read_model {
    int pcr_rate_factor;            /* PCRErrorModel: 1 HOSTILE, 2 AGGRESSIVE, 3 CONSERVATIVE; 0 = off */
    int base_quality_threshold;     /* BASE_QUALITY_SCORE_THRESHOLD, 18 */
    int constant_gcp;               /* gcpHMM, 10; < 0 keeps in->gcp */
    double log10_mismapping_rate;   /* -4.5 (phredScaledGlobalReadMismappingRate 45); -inf = off */
    double max_error_per_base;      /* EXPECTED_ERROR_RATE_PER_BASE, 0.02 */
} mgx_read_model_t;
void mgx_read_model_defaults(mgx_read_model_t* m);
int mgx_pairhmm_region(mgx_pairhmm_t* ctx, const mgx_pairhmm_input_t* in, const uint8_t* mapq,
                       const mgx_read_model_t* model, double* out_log10, uint8_t* out_keep);

/* Rows F1 + F2 together: the same for several regions (samples, active regions) in one device batch.
 * mapq[g], out_log10[g], out_keep[g] belong to region g; out_keep and its entries may be NULL. */
int mgx_pairhmm_regions(mgx_pairhmm_t* ctx, uint32_t n_regions, const mgx_pairhmm_input_t* regions,
                        const uint8_t* const* mapq, const mgx_read_model_t* model,
                        double* const* out_log10, uint8_t* const* out_keep);


/* ---- Host work queue (BASELINE.json configs[2]: one long stream of test cases, cut into batches that
 * worker threads pull off an atomic index -- the reference's threadFunc / atomic region index,
 * main.cpp:254, and its tail-phase work sharing, main.cpp:302-315, IntelPairHmm.cc:296-330).
 * A lane is a host thread with its own context (streams, recycled pinned slabs): it packs the next batch
 * (only the reads / haplotypes its test cases reference cross PCIe, each once), uploads, launches, and
 * collects a batch's results `depth` batches later, so packing and H2D of batch k+1 overlap the kernels
 * of batch k.  Lanes of all devices share one counter; there is no collective and no device-to-device
 * traffic.  Results are identical to mgx_pairhmm_compute on the whole stream.  A queue runs one stream at a time
 * (its run functions are not re-entrant); the caller's thread works as lane 0, the other lanes are threads the
 * run starts and joins. */
typedef struct mgx_pairhmm_queue mgx_pairhmm_queue_t;
typedef struct mgx_pairhmm_queue_config {
    uint32_t n_devices;         /* 0 = one device, ordinal 0 */
    const int* devices;         /* [n_devices] HIP ordinals (an ordinal may repeat) */
    uint32_t lanes_per_device;  /* host threads per device; 0 = 4 */
    uint32_t depth;             /* batches in flight per lane; 0 = 2 */
    uint32_t batch_pairs;       /* test cases per batch; 0 = 65536 */
    unsigned flags;             /* as mgx_pairhmm_create */
} mgx_pairhmm_queue_config_t;
typedef struct mgx_pairhmm_queue_stats {
    uint64_t n_pairs, n_batches, cells;
    uint64_t bytes_h2d, bytes_d2h;      /* what crossed PCIe */
    double seconds;                     /* wall time of the last run, host buffers in -> results in host memory */
    double pack_seconds, wait_seconds;  /* summed over lanes: planning + packing | blocked on the device */
    uint32_t n_lanes;
    uint64_t batches_per_device[16];
} mgx_pairhmm_queue_stats_t;
int mgx_pairhmm_queue_create(const mgx_pairhmm_queue_config_t* cfg, mgx_pairhmm_queue_t** out);
void mgx_pairhmm_queue_destroy(mgx_pairhmm_queue_t* q);
/* Synchronous: the whole stream `in` (pair list, or cross-product form enumerated read-major) in,
 * out_log10[i] for test case i out; used_f64 may be NULL. */
int mgx_pairhmm_queue_run(mgx_pairhmm_queue_t* q, const mgx_pairhmm_input_t* in, double* out_log10, uint8_t* used_f64);
/* The same for test cases [pair_begin, pair_end) only -- the shard of one process when the stream is
 * split over several (one process per GPU): out_log10[i - pair_begin]. */
int mgx_pairhmm_queue_run_range(mgx_pairhmm_queue_t* q, const mgx_pairhmm_input_t* in, uint64_t pair_begin, uint64_t pair_end,
                                double* out_log10, uint8_t* used_f64);
/* Row F1 through the queue: many active regions in the cross-product form (what HipLoglessPairHMM::enqueue parks);
 * lanes pull runs of whole regions holding about batch_pairs test cases, so that flattening and uploading the next
 * regions overlaps the kernels of the previous ones.  out_log10[g] receives region g's [n_reads][n_haps] block;
 * values are identical to mgx_pairhmm_compute_regions / one mgx_pairhmm_compute per region. */
int mgx_pairhmm_queue_run_regions(mgx_pairhmm_queue_t* q, uint32_t n_regions, const mgx_pairhmm_input_t* regions,
                                  double* const* out_log10);
int mgx_pairhmm_queue_stats(mgx_pairhmm_queue_t* q, mgx_pairhmm_queue_stats_t* out);
/* Host only (no device): the queue's packer.  Test cases [pair_begin, pair_end) of `in` as a self-contained
 * batch laid out in buf: *out points into buf, local indices, every referenced read / haplotype once, in
 * first-use order.  *need receives the bytes required; -ENOSPC if buf_bytes is smaller (buf may be NULL). */
int mgx_pairhmm_pack_batch(const mgx_pairhmm_input_t* in, uint64_t pair_begin, uint64_t pair_end, void* buf, size_t buf_bytes,
                           mgx_pairhmm_input_t* out, size_t* need);

/* The two probability tables as built by the product (for table-parity tests):
 * which = 0: ph2pr[128]; which = 1: matchToMatchProb[32640].  Returns the element count. */
int mgx_pairhmm_table_f32(int which, const float** out);
int mgx_pairhmm_table_f64(int which, const double** out);

#ifdef __cplusplus
}
#endif
#endif /* MGX_PAIRHMM_H */
