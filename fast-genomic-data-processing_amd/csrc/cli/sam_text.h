// sam_text.h -- minimal SAM text reader for the sortmardup-compatible CLI (fresh code, no htslib).
//
// Covers what sortmardup needs from htslib's sam_hdr_read / sam_parse1 (the reference parses every
// line with sam_parse1, sortmardup/tbb/bam_parser.cpp:46): the 11 mandatory fields plus the
// optional tags, kept in BAM binary form so a record can be written out unchanged.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace samtext {

struct Header {
    std::string text;                       // all '@' lines, verbatim
    std::vector<std::string> ref_name;
    std::vector<uint64_t> ref_len;
    int find(const char* name, size_t len) const;
};

// One alignment in BAM-ready form.
struct Record {
    int32_t tid = -1, pos = -1, mtid = -1, mpos = -1, tlen = 0;
    uint16_t flag = 0;
    uint8_t mapq = 0;
    uint32_t l_seq = 0;
    std::string qname;
    std::vector<uint32_t> cigar;            // len << 4 | op
    std::vector<uint8_t> seq4;              // 4-bit packed bases, (l_seq + 1) / 2 bytes
    std::vector<uint8_t> qual;              // phred, l_seq bytes (0xFF when '*')
    std::vector<uint8_t> aux;               // BAM-encoded optional fields
    int32_t end() const;                    // 0-based exclusive end on the reference (pos + 1 if no ref length)
};

// Parses header lines out of `data` (stops at the first line not starting with '@');
// returns the offset of the first alignment line.
size_t parse_header(const char* data, size_t size, Header* h);

// Parses one alignment line [line, line + len) (no trailing newline).  Returns false and sets *err.
bool parse_record(const char* line, size_t len, const Header& h, Record* out, std::string* err);

}  // namespace samtext
