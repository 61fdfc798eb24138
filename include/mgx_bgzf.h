/*
 * mgx_bgzf.h -- C ABI of the MI355X BGZF block compressor (libmgx.so), SURVEY.md section 8f row F3.
 *
 * Drop-in seam: the reference's writer threads hand every 64 KB of BAM bytes to htslib's
 * bgzf_compress (deepmutect/htslib/bgzf.c:610-648, called from deflate_block, bgzf.c:695-707, by
 * sortmardup/main.cpp:371-421), which runs zlib's deflate on one CPU thread per slice -- ~80 % of the
 * wall time of a coordinate sort on this machine.  Here a batch of blocks is compressed on the device:
 * one workgroup per BGZF block (LZ77 matching through an LDS hash table, greedy/lazy parse, dynamic
 * Huffman codes built per block, CRC-32), and the finished blocks come back packed back to back.
 *
 * Every block is a complete gzip member with the BGZF extra field (SAMv1 section 4.1): any inflater
 * returns the input bytes.  The compressed BYTES are this implementation's own (as they are zlib's own in
 * the reference, and depend on its level and version); what is identical to the reference is the
 * uncompressed stream.
 *
 * All functions return 0 or a negative errno-style code; mgx_last_error() has the message.
 */
#ifndef MGX_BGZF_H
#define MGX_BGZF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mgx_bgzf mgx_bgzf_t;
typedef struct mgx_bgzf_batch mgx_bgzf_batch_t;

#define MGX_BGZF_MAX_BLOCK_IN 0xff00u      /* uncompressed bytes per block (htslib's BGZF_BLOCK_SIZE) */
#define MGX_BGZF_MAX_BLOCK_OUT 0x10000u    /* upper bound of one finished block */

const char* mgx_last_error(void);

int mgx_bgzf_create(int device, unsigned flags, mgx_bgzf_t** out);
void mgx_bgzf_destroy(mgx_bgzf_t* ctx);
/* Optional: sets up the compressor's device state (scratch, kernel attributes) now instead of at the first batch, e.g. on
 * a bring-up thread while the caller parses its input.  mgx_bgzf_create itself only starts the runtime and two streams. */
int mgx_bgzf_prepare(mgx_bgzf_t* ctx);
/* Free and total device memory, for a caller that has to choose between keeping its records in HBM and in host memory. */
int mgx_bgzf_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes);

/* A batch owns a pinned input buffer the caller fills (uncompressed bytes of consecutive blocks and their
 * n_blocks + 1 offsets), device memory, and a pinned output buffer.  Several batches may be in flight on
 * one context: submit() only enqueues (H2D, kernels, D2H of the offsets and of the packed blocks). */
int mgx_bgzf_batch_create(mgx_bgzf_t* ctx, uint64_t in_capacity, uint32_t max_blocks, mgx_bgzf_batch_t** out);
void mgx_bgzf_batch_destroy(mgx_bgzf_t* ctx, mgx_bgzf_batch_t* b);
uint8_t* mgx_bgzf_batch_input(mgx_bgzf_batch_t* b);        /* [in_capacity] */
uint64_t* mgx_bgzf_batch_offsets(mgx_bgzf_batch_t* b);     /* [max_blocks + 1], offsets[0] = 0, block i = [offsets[i], offsets[i+1]) */
/* offsets[i+1] - offsets[i] <= MGX_BGZF_MAX_BLOCK_IN (an empty block is allowed and yields an empty gzip member) */
int mgx_bgzf_batch_submit(mgx_bgzf_t* ctx, mgx_bgzf_batch_t* b, uint32_t n_blocks);
/* Waits for the batch.  *out = the finished blocks back to back (pinned memory owned by the batch, valid until
 * the next submit), out_offsets[i] = start of block i in it, out_offsets[n_blocks] = total bytes. */
int mgx_bgzf_batch_wait(mgx_bgzf_t* ctx, mgx_bgzf_batch_t* b, const uint8_t** out, const uint64_t** out_offsets);

/* One shot over pageable memory: pieces [offsets[i], offsets[i+1]) of `in` -> BGZF blocks appended to `out`
 * (capacity out_capacity >= mgx_bgzf_bound(n_bytes, n_blocks)); out_offsets has n_blocks + 1 entries. */
uint64_t mgx_bgzf_bound(uint64_t n_bytes, uint64_t n_blocks);
int mgx_bgzf_compress(mgx_bgzf_t* ctx, const uint8_t* in, const uint64_t* offsets, uint64_t n_blocks, uint8_t* out,
                      uint64_t out_capacity, uint64_t* out_offsets);

/* ---- Record store: the BAM records live in HBM from ingest to output ------------------------------------------------
 * What the reference keeps on disk between its passes (LZ4-compressed range partitions, sortmardup/main.cpp:194-227) and
 * reads back record by record in sorted order (main.cpp:385-397) stays on the device here: put() copies a parser's
 * records (BAM alignment records WITHOUT their block_size field, any number per call, from any thread) into HBM and says
 * where they are; emit() writes the stream "block_size, record" for q = 0 .. n-1 in the order given -- FLAG |= 0x400 where
 * the record is marked duplicate (main.cpp:385-388) --, cuts it every 65 280 bytes (records may span blocks, SAMv1 4.1),
 * compresses the blocks and hands them to `sink` in order.  uoff[q] (n + 1 entries) is record q's offset in the
 * uncompressed stream: together with the compressed offsets of the blocks it gives the virtual offsets of the index. */
typedef struct mgx_bgzf_store mgx_bgzf_store_t;
typedef int (*mgx_bgzf_sink_t)(void* user, const uint8_t* blocks, uint64_t n_bytes, uint32_t n_blocks, const uint64_t* block_offsets);
int mgx_bgzf_store_create(mgx_bgzf_t* ctx, mgx_bgzf_store_t** out);
void mgx_bgzf_store_destroy(mgx_bgzf_store_t* st);
/* Optional: allocate HBM for about `bytes` of records now (at most one 256 MB piece) instead of inside the first put. */
int mgx_bgzf_store_reserve(mgx_bgzf_store_t* st, uint64_t bytes);
int mgx_bgzf_store_put(mgx_bgzf_store_t* st, const uint8_t* bytes, uint64_t n_bytes, uint64_t* device_address);
/* order[q] = index (into addr / len / dup) of the q-th record of the output; addr[i] = device address of record i
 * (put()'s address plus the record's offset in that call's bytes), len[i] its length */
int mgx_bgzf_store_emit(mgx_bgzf_store_t* st, uint64_t n, const uint32_t* order, const uint8_t* dup, const uint64_t* addr,
                        const uint32_t* len, mgx_bgzf_sink_t sink, void* user, uint64_t* uoff);

typedef struct mgx_bgzf_stats {
    uint64_t n_blocks, bytes_in, bytes_out;   /* since create */
    uint64_t n_stored;                        /* blocks emitted as stored (incompressible) */
    float ms_kernels;                         /* last batch waited for: the deflate kernel (HIP events on its stream) */
    float ms_pack;                            /* ... and offsets + pack on the copy stream: the pack kernel stores the finished blocks
                                               * straight into the pinned output buffer, i.e. this is the device-to-host transfer */
} mgx_bgzf_stats_t;
int mgx_bgzf_stats(mgx_bgzf_t* ctx, mgx_bgzf_stats_t* out);

#ifdef __cplusplus
}
#endif
#endif
