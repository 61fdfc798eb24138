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

// What the sort / mark-duplicate key derivation and the index need of one record, next to its BAM bytes.
struct Parsed {
    uint16_t flag; int32_t tid, pos, end;   // end: 0-based exclusive end on the reference (pos + 1 if no ref length)
    uint16_t score;                         // BAMRecord::score (sortmardup/tbb/bam_record.cpp:7-14): qualities of at least 15, summed in 16 bits
};
// The same parse in one pass without a Record: the BAM encoding of the line (what bamout::encode_record makes of
// parse_record's result, byte for byte) is APPENDED to *blob, its CIGAR operations / phred qualities / read name to
// *cigar / *qual / *qname (qual may be NULL: the qualities are on the record, and their score in *out).  On failure nothing stays
// appended and *err says why.
bool parse_record_into(const char* line, size_t len, const Header& h, Parsed* out, std::vector<uint32_t>* cigar, std::vector<uint8_t>* qual,
                       std::vector<char>* qname, std::vector<uint8_t>* blob, std::string* err);

}  // namespace samtext

namespace bamout {
// BAM-encodes one record WITHOUT the leading block_size field (so the flag sits at byte 14).
void encode_record(const samtext::Record& r, std::vector<uint8_t>* out);
constexpr size_t kFlagOffset = 14;
int reg2bin(int64_t beg, int64_t end);
}  // namespace bamout
