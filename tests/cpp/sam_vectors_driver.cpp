// tests/cpp/sam_vectors_driver.cpp -- test driver (not product code): runs the CLI's SAM text parser
// (fast-genomic-data-processing_amd/csrc/cli/sam_text.cpp) over a whole SAM file and returns the BAM encoding of every
// alignment line, block_size-prefixed, as the writer would put them into the stream.  Both parse paths are run -- the
// one-pass parse_record_into() the ingest uses and parse_record() + bamout::encode_record() -- and must agree byte for byte.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "sam_text.h"

extern "C" int sam_text_to_bam(const char* text, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* n_out, uint32_t* n_rec,
                               uint32_t* n_ref, char* err, int err_cap) {
    auto fail = [&](const std::string& why, uint32_t line_no) {
        if (err && err_cap > 0) { std::string m = "line " + std::to_string(line_no) + ": " + why; strncpy(err, m.c_str(), (size_t)err_cap - 1); err[err_cap - 1] = 0; }
        return -1;
    };
    samtext::Header h;
    size_t p = samtext::parse_header(text, (size_t)n, &h);
    *n_ref = (uint32_t)h.ref_name.size();
    std::vector<uint8_t> stream, blob, blob2, qual;
    std::vector<uint32_t> cigar;
    std::vector<char> qname;
    uint32_t recs = 0, line_no = 0;
    while (p < n) {
        const char* nl = (const char*)memchr(text + p, '\n', (size_t)n - p);
        size_t len = nl ? (size_t)(nl - (text + p)) : (size_t)n - p;
        const char* line = text + p;
        p += len + 1;
        ++line_no;
        if (len && line[len - 1] == '\r') --len;
        if (len == 0) continue;
        std::string e;
        samtext::Parsed ps;
        blob.clear(); cigar.clear(); qual.clear(); qname.clear();
        if (!samtext::parse_record_into(line, len, h, &ps, &cigar, &qual, &qname, &blob, &e)) return fail(e, line_no);
        samtext::Record r;
        if (!samtext::parse_record(line, len, h, &r, &e)) return fail("parse_record: " + e, line_no);
        blob2.clear();
        bamout::encode_record(r, &blob2);
        if (blob != blob2) return fail("the two parse paths disagree", line_no);
        if (ps.flag != r.flag || ps.tid != r.tid || ps.pos != r.pos || ps.end != r.end()) return fail("Parsed summary differs from the record", line_no);
        const uint32_t bs = (uint32_t)blob.size();
        const uint8_t* b = (const uint8_t*)&bs;
        stream.insert(stream.end(), b, b + 4);
        stream.insert(stream.end(), blob.begin(), blob.end());
        ++recs;
    }
    *n_out = stream.size(); *n_rec = recs;
    if (stream.size() > cap) return fail("output buffer too small", 0);
    if (!stream.empty()) memcpy(out, stream.data(), stream.size());
    return 0;
}
