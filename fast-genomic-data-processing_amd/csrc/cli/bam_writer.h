// bam_writer.h -- BGZF / BAM / BAI output for the sortmardup-compatible CLI (fresh code on zlib).
//
// Replaces what sortmardup/main.cpp:359-465 does with a patched htslib (bam_write_idx2,
// bgzf_flush2, hts_close2, merge_index, hts_idx_finish3 -- functions that exist nowhere in the
// reference tree): per-thread compression of contiguous slices of the sorted records into
// independent BGZF blocks, concatenation, and a BAI index built from the records' virtual offsets.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "sam_text.h"

namespace bamout {

// (encode_record, kFlagOffset and reg2bin live next to the parser: sam_text.h)

struct RecordRef {              // one record of the output, in output order
    const uint8_t* blob;        // encode_record() bytes
    uint32_t len;
    int32_t tid, beg, end;      // for the index (end exclusive)
    bool set_dup;               // OR 0x400 into the flag while writing
    bool mapped;                // !(flag & 4), for the index metadata
};

// Writes <path> and <path>.bai.  Returns false and sets *err on I/O failure.
// device >= 0: the BGZF blocks are compressed on that HIP device (include/mgx_bgzf.h), the writer threads only gather
// the records; device < 0: zlib at `level` on the writer threads (the reference's way, bgzf.c:610).
bool write_bam(const std::string& path, const samtext::Header& hdr, const std::vector<RecordRef>& recs,
               int threads, int level, int device, std::string* err);

// The same with the records resident in HBM: recs[k].blob is the DEVICE address mgx_bgzf_store_put() returned for the
// record; bgzf_ctx / store are the mgx_bgzf_t* / mgx_bgzf_store_t* that hold them (include/mgx_bgzf.h).  The device gathers
// the records in output order, sets the duplicate flags, cuts and compresses the stream; the host writes blocks and index.
bool write_bam_store(const std::string& path, const samtext::Header& hdr, const std::vector<RecordRef>& recs, void* bgzf_ctx, void* store,
                     int threads, std::string* err);

}  // namespace bamout
