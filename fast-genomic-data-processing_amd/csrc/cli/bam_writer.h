// bam_writer.h -- BGZF / BAM / BAI output for the sortmardup-compatible CLI (fresh code on zlib).
//
// Replaces what sortmardup/main.cpp:359-465 does with a patched htslib (bam_write_idx2,
// bgzf_flush2, hts_close2, merge_index, hts_idx_finish3 -- functions that exist nowhere in the
// reference tree): per-thread compression of contiguous slices of the sorted records into
// independent BGZF blocks, concatenation, and a BAI index built from the records' virtual offsets.
#pragma once

#include <atomic>
#include <cstdint>
#include <memory>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "sam_text.h"

namespace bamout {

// (encode_record, kFlagOffset and reg2bin live next to the parser: sam_text.h)

// std::vector whose resize() / sized constructor leaves trivial elements uninitialised: an array of 200 M records is several GB,
// and value-initialising it is one thread writing zeros to all of it (and touching every page first) before the threads that
// fill it start -- 1-2 s per array at BASELINE configs[3]
template <class T>
struct default_init_allocator : std::allocator<T> {
    template <class U> struct rebind { using other = default_init_allocator<U>; };
    default_init_allocator() = default;
    template <class U> default_init_allocator(const default_init_allocator<U>&) {}
    template <class U> void construct(U* p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void*>(p)) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <class T> using NoInitVector = std::vector<T, default_init_allocator<T>>;

struct RecordRef {              // one record of the output, in output order
    const uint8_t* blob;        // encode_record() bytes
    uint32_t len;
    int32_t tid, beg, end;      // for the index (end exclusive)
    bool set_dup;               // OR 0x400 into the flag while writing
    bool mapped;                // !(flag & 4), for the index metadata
};

using RecordRefs = NoInitVector<RecordRef>;

// set to 1 by write_bam_store once its per-record arrays are built (the caller's clean-up thread waits for that: page frees and
// first touches of 5 GB of new arrays slow each other down)
extern std::atomic<int> g_store_arrays_ready;

// Writes <path> and <path>.bai.  Returns false and sets *err on I/O failure.
// device >= 0: the BGZF blocks are compressed on that HIP device (include/mgx_bgzf.h), the writer threads only gather
// the records; device < 0: zlib at `level` on the writer threads (the reference's way, bgzf.c:610).
bool write_bam(const std::string& path, const samtext::Header& hdr, const RecordRefs& recs,
               int threads, int level, int device, std::string* err);

// The same with the records resident in HBM: recs[k].blob is the DEVICE address mgx_bgzf_store_put() returned for the
// record; bgzf_ctx / store are the mgx_bgzf_t* / mgx_bgzf_store_t* that hold them (include/mgx_bgzf.h).  The device gathers
// the records in output order, sets the duplicate flags, cuts and compresses the stream; the host writes blocks and index.
bool write_bam_store(const std::string& path, const samtext::Header& hdr, const RecordRefs& recs, void* bgzf_ctx, void* store,
                     int threads, std::string* err);

}  // namespace bamout
