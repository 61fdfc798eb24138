// sam_text.cpp -- see sam_text.h.  Follows the SAM/BAM specification (SAMv1 sections 1.4, 4.2).
#include "sam_text.h"

#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <limits>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace samtext {

int Header::find(const char* name, size_t len) const {
    for (size_t i = 0; i < ref_name.size(); ++i)
        if (ref_name[i].size() == len && memcmp(ref_name[i].data(), name, len) == 0) return (int)i;
    return -1;
}

int32_t Record::end() const {
    int64_t ref = 0;
    for (uint32_t c : cigar) {
        const uint32_t op = c & 15;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref += c >> 4;   // M D N = X
    }
    return (int32_t)(pos + (ref > 0 ? ref : 1));
}

size_t parse_header(const char* data, size_t size, Header* h) {
    size_t off = 0;
    while (off < size && data[off] == '@') {
        const char* nl = (const char*)memchr(data + off, '\n', size - off);
        const size_t len = nl ? (size_t)(nl - (data + off)) : size - off;
        const char* line = data + off;
        if (len >= 3 && line[1] == 'S' && line[2] == 'Q') {
            std::string name; uint64_t ln = 0;
            size_t i = 3;
            while (i < len) {
                if (line[i] == '\t') { ++i; continue; }
                size_t j = i;
                while (j < len && line[j] != '\t') ++j;
                if (j - i >= 3 && line[i + 2] == ':') {
                    if (line[i] == 'S' && line[i + 1] == 'N') name.assign(line + i + 3, j - i - 3);
                    else if (line[i] == 'L' && line[i + 1] == 'N') ln = strtoull(std::string(line + i + 3, j - i - 3).c_str(), nullptr, 10);
                }
                i = j;
            }
            h->ref_name.push_back(name);
            h->ref_len.push_back(ln);
        }
        h->text.append(line, len);
        h->text.push_back('\n');
        off += len + (nl ? 1 : 0);
    }
    return off;
}

namespace {

const uint8_t* nt16_table() {
    static uint8_t t[256];
    static bool init = false;
    if (!init) {
        memset(t, 15, sizeof t);
        const char* codes = "=ACMGRSVTWYHKDBN";
        for (int i = 0; i < 16; ++i) { t[(uint8_t)codes[i]] = (uint8_t)i; t[(uint8_t)(codes[i] | 0x20)] = (uint8_t)i; }
        init = true;
    }
    return t;
}

// the first tab in [p, end), or end.  Most fields of a SAM line are a few bytes long: a call of memchr per field costs more than
// the scan (round 3: the parse was ~180 ns of a record's ~700 ns on a parser thread)
inline const char* find_tab(const char* p, const char* end) {
#if defined(__SSE2__)
    const __m128i tab = _mm_set1_epi8('\t');
    while (p + 16 <= end) {
        const int m = _mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(p)), tab));
        if (m) return p + __builtin_ctz((unsigned)m);
        p += 16;
    }
#endif
    while (p < end && *p != '\t') ++p;
    return p;
}

inline bool next_field(const char*& p, const char* end, const char** f, size_t* n) {
    if (p > end) return false;
    const char* t = find_tab(p, end);
    *f = p;
    *n = (size_t)(t - p);
    p = t + 1;                                        // one past the tab, or end + 1 when the line ends here
    return true;
}

// 4-bit codes of l bases into (l + 1) / 2 bytes.  Sixteen bases at a time where they are all A C G T N (either case) and the CPU
// has SSSE3: two table look-ups by the ASCII code's low nibble (the code, and the letter that nibble stands for, which must be
// the base itself), then pairs folded with one multiply-add; anything else takes the 256-entry table, byte by byte.
#if defined(__x86_64__)
__attribute__((target("ssse3"))) size_t pack_bases_ssse3(const uint8_t* sq, size_t l, uint8_t* w) {
    const __m128i code = _mm_setr_epi8(-1, 1, -1, 2, 8, -1, -1, 4, -1, -1, -1, -1, -1, -1, 15, -1);
    const __m128i letter = _mm_setr_epi8(-1, 'A', -1, 'C', 'T', -1, -1, 'G', -1, -1, -1, -1, -1, -1, 'N', -1);
    const __m128i low = _mm_set1_epi8(0x0F), upper = _mm_set1_epi8((char)0xDF), weight = _mm_set1_epi16(0x0110);
    size_t i = 0;
    for (; i + 16 <= l; i += 16) {
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(sq + i));
        const __m128i nib = _mm_and_si128(b, low);
        if (_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_and_si128(b, upper), _mm_shuffle_epi8(letter, nib))) != 0xFFFF) break;
        const __m128i pairs = _mm_maddubs_epi16(_mm_shuffle_epi8(code, nib), weight);          // code[2k] * 16 + code[2k + 1]
        _mm_storel_epi64(reinterpret_cast<__m128i*>(w + i / 2), _mm_packus_epi16(pairs, pairs));
    }
    return i;                                         // bases done (a multiple of 16)
}
#endif
inline void pack_bases(const uint8_t* sq, size_t l, uint8_t* w) {
    size_t i = 0;
#if defined(__x86_64__)
    static const bool ssse3 = __builtin_cpu_supports("ssse3");
    if (ssse3 && l >= 16) i = pack_bases_ssse3(sq, l, w);
#endif
    const uint8_t* nt16 = nt16_table();
    size_t k = i / 2;
    for (; 2 * k + 1 < l; ++k) w[k] = (uint8_t)(nt16[sq[2 * k]] << 4 | nt16[sq[2 * k + 1]]);
    if (l & 1) w[k] = (uint8_t)(nt16[sq[l - 1]] << 4);
}

// phred values of l quality characters into q, and what BAMRecord::score makes of them (sortmardup/tbb/bam_record.cpp:7-14: the
// sum of the values of at least 15 in a uint16_t, wrapping)
inline uint16_t phred_and_score(const char* src, size_t l, uint8_t* q) {
    size_t i = 0;
    uint64_t sum = 0;
#if defined(__SSE2__)
    const __m128i off = _mm_set1_epi8(33), min15 = _mm_set1_epi8(15), zero = _mm_setzero_si128();
    __m128i acc = zero;
    for (; i + 16 <= l; i += 16) {
        const __m128i v = _mm_sub_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i)), off);
        _mm_storeu_si128(reinterpret_cast<__m128i*>(q + i), v);
        const __m128i keep = _mm_cmpeq_epi8(_mm_max_epu8(v, min15), v);                        // v >= 15, unsigned
        acc = _mm_add_epi64(acc, _mm_sad_epu8(_mm_and_si128(v, keep), zero));
    }
    sum = (uint64_t)_mm_cvtsi128_si64(acc) + (uint64_t)_mm_cvtsi128_si64(_mm_unpackhi_epi64(acc, acc));
#endif
    for (; i < l; ++i) { const uint8_t v = (uint8_t)(src[i] - 33); q[i] = v; sum += v >= 15 ? v : 0; }
    return (uint16_t)sum;
}

// decimal integer with an optional sign, nothing else in the field (what strtoll accepts, minus leading blanks)
inline bool to_int(const char* f, size_t n, int64_t* v) {
    if (n == 0 || n > 20) return false;
    size_t i = 0;
    const bool neg = f[0] == '-';
    if (f[0] == '-' || f[0] == '+') i = 1;
    if (i == n || n - i > 18) {                      // no digit, or close to overflow: the careful way
        if (i == n) return false;
        char buf[24];
        memcpy(buf, f, n); buf[n] = 0;
        char* e; errno = 0;
        const long long x = strtoll(buf, &e, 10);
        if (errno || *e) return false;
        *v = x;
        return true;
    }
    int64_t x = 0;
    for (; i < n; ++i) {
        const unsigned d = (unsigned)(f[i] - '0');
        if (d > 9) return false;
        x = x * 10 + (int64_t)d;
    }
    *v = neg ? -x : x;
    return true;
}

template <typename T>
inline void put(std::vector<uint8_t>& a, T v) { const uint8_t* p = (const uint8_t*)&v; a.insert(a.end(), p, p + sizeof(T)); }

// smallest BAM integer type holding v (what htslib chooses when it parses an 'i' tag)
void put_int_tag(std::vector<uint8_t>& a, int64_t v) {
    if (v < 0) {
        if (v >= -128) { a.push_back('c'); put<int8_t>(a, (int8_t)v); }
        else if (v >= -32768) { a.push_back('s'); put<int16_t>(a, (int16_t)v); }
        else { a.push_back('i'); put<int32_t>(a, (int32_t)v); }
    } else {
        if (v <= 255) { a.push_back('C'); put<uint8_t>(a, (uint8_t)v); }
        else if (v <= 65535) { a.push_back('S'); put<uint16_t>(a, (uint16_t)v); }
        else { a.push_back('I'); put<uint32_t>(a, (uint32_t)v); }
    }
}

bool parse_aux(const char* f, size_t n, std::vector<uint8_t>& a, std::string* err) {
    if (n < 5 || f[2] != ':' || f[4] != ':') { *err = "malformed optional field"; return false; }
    a.push_back((uint8_t)f[0]); a.push_back((uint8_t)f[1]);
    const char type = f[3];
    const char* v = f + 5; const size_t vn = n - 5;
    switch (type) {
        case 'A': a.push_back('A'); a.push_back(vn ? (uint8_t)v[0] : 0); return true;
        case 'i': { int64_t x; if (!to_int(v, vn, &x)) { *err = "bad integer tag"; return false; } put_int_tag(a, x); return true; }
        case 'f': a.push_back('f'); put<float>(a, strtof(std::string(v, vn).c_str(), nullptr)); return true;
        case 'Z': case 'H': a.push_back((uint8_t)type); a.insert(a.end(), v, v + vn); a.push_back(0); return true;
        case 'B': {
            if (vn < 1) { *err = "bad B tag"; return false; }
            const char sub = v[0];
            std::vector<std::string> items;
            size_t i = 1;
            while (i < vn) { if (v[i] == ',') { ++i; continue; } size_t j = i; while (j < vn && v[j] != ',') ++j; items.emplace_back(v + i, j - i); i = j; }
            a.push_back('B'); a.push_back((uint8_t)sub); put<uint32_t>(a, (uint32_t)items.size());
            for (auto& it : items) {
                switch (sub) {
                    case 'c': put<int8_t>(a, (int8_t)strtol(it.c_str(), nullptr, 10)); break;
                    case 'C': put<uint8_t>(a, (uint8_t)strtoul(it.c_str(), nullptr, 10)); break;
                    case 's': put<int16_t>(a, (int16_t)strtol(it.c_str(), nullptr, 10)); break;
                    case 'S': put<uint16_t>(a, (uint16_t)strtoul(it.c_str(), nullptr, 10)); break;
                    case 'i': put<int32_t>(a, (int32_t)strtol(it.c_str(), nullptr, 10)); break;
                    case 'I': put<uint32_t>(a, (uint32_t)strtoul(it.c_str(), nullptr, 10)); break;
                    case 'f': put<float>(a, strtof(it.c_str(), nullptr)); break;
                    default: *err = "bad B subtype"; return false;
                }
            }
            return true;
        }
        default: *err = "unknown optional field type"; return false;
    }
}

}  // namespace

namespace {
// reference name -> index; neighbouring records name the same sequence, so the last hit is tried first
inline int find_ref(const Header& h, const char* name, size_t len) {
    static thread_local int last = -1;
    if (last >= 0 && (size_t)last < h.ref_name.size() && h.ref_name[last].size() == len && memcmp(h.ref_name[last].data(), name, len) == 0) return last;
    const int k = h.find(name, len);
    if (k >= 0) last = k;
    return k;
}
}  // namespace

bool parse_record(const char* line, size_t len, const Header& h, Record* r, std::string* err) {
    const char* p = line; const char* end = line + len;
    const char* f[11]; size_t n[11];
    for (int k = 0; k < 11; ++k)
        if (!next_field(p, end, &f[k], &n[k])) { *err = "fewer than 11 fields"; return false; }
    int64_t v;
    r->qname.assign(f[0], n[0]);
    if (n[0] == 0 || n[0] > 254) { *err = "bad QNAME length"; return false; }
    if (!to_int(f[1], n[1], &v) || v < 0 || v > 65535) { *err = "bad FLAG"; return false; }
    r->flag = (uint16_t)v;
    r->tid = (n[2] == 1 && f[2][0] == '*') ? -1 : find_ref(h, f[2], n[2]);
    if (r->tid < 0 && !(n[2] == 1 && f[2][0] == '*')) { *err = "RNAME not in the header"; return false; }
    if (!to_int(f[3], n[3], &v)) { *err = "bad POS"; return false; }
    r->pos = (int32_t)v - 1;
    if (!to_int(f[4], n[4], &v) || v < 0 || v > 255) { *err = "bad MAPQ"; return false; }
    r->mapq = (uint8_t)v;
    r->cigar.clear();
    if (!(n[5] == 1 && f[5][0] == '*')) {
        uint64_t num = 0; bool have = false;
        for (size_t i = 0; i < n[5]; ++i) {
            const char ch = f[5][i];
            if (ch >= '0' && ch <= '9') { num = num * 10 + (uint64_t)(ch - '0'); have = true; continue; }
            const char* ops = "MIDNSHP=XB";
            const char* q = strchr(ops, ch);
            if (!q || !have || num >= (1ull << 28)) { *err = "bad CIGAR"; return false; }
            r->cigar.push_back((uint32_t)(num << 4) | (uint32_t)(q - ops));
            num = 0; have = false;
        }
        if (have) { *err = "bad CIGAR"; return false; }
        // n_cigar_op is 16 bits in BAM; htslib writes longer CIGARs as a kSmN placeholder plus a CG:B,I tag (SAMv1 4.2.2),
        // which this tool does not produce: reject instead of writing a record every reader would mis-slice
        if (r->cigar.size() > 65535) { *err = "more than 65535 CIGAR operations"; return false; }
    }
    if (n[6] == 1 && f[6][0] == '=') r->mtid = r->tid;
    else if (n[6] == 1 && f[6][0] == '*') r->mtid = -1;
    else { r->mtid = find_ref(h, f[6], n[6]); if (r->mtid < 0) { *err = "RNEXT not in the header"; return false; } }
    if (!to_int(f[7], n[7], &v)) { *err = "bad PNEXT"; return false; }
    r->mpos = (int32_t)v - 1;
    if (!to_int(f[8], n[8], &v)) { *err = "bad TLEN"; return false; }
    r->tlen = (int32_t)v;
    const uint8_t* nt16 = nt16_table();
    if (n[9] == 1 && f[9][0] == '*') { r->l_seq = 0; r->seq4.clear(); }
    else {
        r->l_seq = (uint32_t)n[9];
        r->seq4.resize((n[9] + 1) / 2);
        const uint8_t* sq = (const uint8_t*)f[9];
        size_t k = 0;
        for (; 2 * k + 1 < n[9]; ++k) r->seq4[k] = (uint8_t)(nt16[sq[2 * k]] << 4 | nt16[sq[2 * k + 1]]);
        if (n[9] & 1) r->seq4[k] = (uint8_t)(nt16[sq[n[9] - 1]] << 4);
    }
    if (n[10] == 1 && f[10][0] == '*') r->qual.assign(r->l_seq, 0xFF);
    else {
        if (n[10] != r->l_seq) { *err = "SEQ and QUAL differ in length"; return false; }
        r->qual.resize(r->l_seq);
        for (size_t i = 0; i < n[10]; ++i) r->qual[i] = (uint8_t)(f[10][i] - 33);
    }
    r->aux.clear();
    const char* af; size_t an;
    while (p <= end && next_field(p, end, &af, &an)) {
        if (an == 0) continue;
        if (!parse_aux(af, an, r->aux, err)) return false;
    }
    return true;
}

bool parse_record_into(const char* line, size_t len, const Header& h, Parsed* out, std::vector<uint32_t>* cigar, std::vector<uint8_t>* qual,
                       std::vector<char>* qname, std::vector<uint8_t>* blob, std::string* err) {
    const size_t cigar0 = cigar->size(), qual0 = qual ? qual->size() : 0, qname0 = qname->size(), blob0 = blob->size();
    auto bad = [&](const char* why) { cigar->resize(cigar0); if (qual) qual->resize(qual0); qname->resize(qname0); blob->resize(blob0); *err = why; return false; };
    const char* p = line; const char* end = line + len;
    const char* f[11]; size_t n[11];
    for (int k = 0; k < 11; ++k)
        if (!next_field(p, end, &f[k], &n[k])) return bad("fewer than 11 fields");
    int64_t v;
    if (n[0] == 0 || n[0] > 254) return bad("bad QNAME length");
    if (!to_int(f[1], n[1], &v) || v < 0 || v > 65535) return bad("bad FLAG");
    const uint16_t flag = (uint16_t)v;
    const bool no_ref = n[2] == 1 && f[2][0] == '*';
    const int32_t tid = no_ref ? -1 : find_ref(h, f[2], n[2]);
    if (tid < 0 && !no_ref) return bad("RNAME not in the header");
    if (!to_int(f[3], n[3], &v)) return bad("bad POS");
    const int32_t pos = (int32_t)v - 1;
    if (!to_int(f[4], n[4], &v) || v < 0 || v > 255) return bad("bad MAPQ");
    const uint8_t mapq = (uint8_t)v;
    int64_t ref_len = 0;
    if (!(n[5] == 1 && f[5][0] == '*')) {
        uint64_t num = 0; bool have = false;
        for (size_t i = 0; i < n[5]; ++i) {
            const char ch = f[5][i];
            if (ch >= '0' && ch <= '9') { num = num * 10 + (uint64_t)(ch - '0'); have = true; continue; }
            int op;
            switch (ch) {
                case 'M': op = 0; break; case 'I': op = 1; break; case 'D': op = 2; break; case 'N': op = 3; break; case 'S': op = 4; break;
                case 'H': op = 5; break; case 'P': op = 6; break; case '=': op = 7; break; case 'X': op = 8; break; case 'B': op = 9; break;
                default: op = -1;
            }
            if (op < 0 || !have || num >= (1ull << 28)) return bad("bad CIGAR");
            cigar->push_back((uint32_t)(num << 4) | (uint32_t)op);
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += (int64_t)num;
            num = 0; have = false;
        }
        if (have) return bad("bad CIGAR");
    }
    const size_t n_cig = cigar->size() - cigar0;
    if (n_cig > 65535) return bad("more than 65535 CIGAR operations");      // see parse_record
    const int32_t rec_end = (int32_t)(pos + (ref_len > 0 ? ref_len : 1));
    int32_t mtid;
    if (n[6] == 1 && f[6][0] == '=') mtid = tid;
    else if (n[6] == 1 && f[6][0] == '*') mtid = -1;
    else { mtid = find_ref(h, f[6], n[6]); if (mtid < 0) return bad("RNEXT not in the header"); }
    if (!to_int(f[7], n[7], &v)) return bad("bad PNEXT");
    const int32_t mpos = (int32_t)v - 1;
    if (!to_int(f[8], n[8], &v)) return bad("bad TLEN");
    const int32_t tlen = (int32_t)v;
    const bool no_seq = n[9] == 1 && f[9][0] == '*';
    const uint32_t l_seq = no_seq ? 0u : (uint32_t)n[9];
    const bool no_qual = n[10] == 1 && f[10][0] == '*';
    if (!no_qual && n[10] != l_seq) return bad("SEQ and QUAL differ in length");
    // ---- the record's fixed part, name, CIGAR, bases, qualities: one resize, then plain stores
    const size_t l_qn = n[0] + 1, n_seq4 = (l_seq + 1) / 2;
    const size_t fixed = 32 + l_qn + 4 * n_cig + n_seq4 + l_seq;
    blob->resize(blob0 + fixed);
    uint8_t* w = blob->data() + blob0;
    auto st = [&w](const void* src, size_t k) { if (k) memcpy(w, src, k); w += k; };
    const int32_t l_seq_i = (int32_t)l_seq;
    const uint16_t bin = (uint16_t)bamout::reg2bin(pos, rec_end), nc = (uint16_t)n_cig;
    const uint8_t lq = (uint8_t)l_qn, zero = 0;
    st(&tid, 4); st(&pos, 4); st(&lq, 1); st(&mapq, 1); st(&bin, 2); st(&nc, 2); st(&flag, 2); st(&l_seq_i, 4); st(&mtid, 4); st(&mpos, 4); st(&tlen, 4);
    st(f[0], n[0]); st(&zero, 1);
    if (n_cig) st(cigar->data() + cigar0, 4 * n_cig);
    pack_bases((const uint8_t*)f[9], l_seq, w);
    w += n_seq4;
    // qualities straight onto the record (and, for a caller that wants them apart, copied from there)
    uint16_t score;
    if (no_qual) { if (l_seq) memset(w, 0xFF, l_seq); score = (uint16_t)(255u * l_seq); }
    else score = phred_and_score(f[10], l_seq, w);
    if (qual) qual->insert(qual->end(), w, w + l_seq);
    w += l_seq;
    qname->insert(qname->end(), f[0], f[0] + n[0]);
    // ---- the optional fields straight onto the record
    const char* af; size_t an;
    while (p <= end && next_field(p, end, &af, &an)) {
        if (an == 0) continue;
        if (!parse_aux(af, an, *blob, err)) { const std::string why = *err; return bad(why.c_str()); }
    }
    out->flag = flag; out->tid = tid; out->pos = pos; out->end = rec_end; out->score = score;
    return true;
}

}  // namespace samtext

namespace bamout {

int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

void encode_record(const samtext::Record& r, std::vector<uint8_t>* out) {
    const size_t l_qn = r.qname.size() + 1, n_cig = r.cigar.size();
    const size_t total = 32 + l_qn + 4 * n_cig + r.seq4.size() + r.qual.size() + r.aux.size();
    const size_t at = out->size();
    out->resize(at + total);
    uint8_t* p = out->data() + at;
    auto w = [&p](const void* src, size_t n) { memcpy(p, src, n); p += n; };
    const int32_t tid = r.tid, pos = r.pos, l_seq = (int32_t)r.l_seq, mtid = r.mtid, mpos = r.mpos, tlen = r.tlen;
    const uint16_t bin = (uint16_t)reg2bin(r.pos, r.end()), nc = (uint16_t)n_cig, flag = r.flag;
    const uint8_t lq = (uint8_t)l_qn, mapq = r.mapq, zero = 0;
    w(&tid, 4); w(&pos, 4); w(&lq, 1); w(&mapq, 1); w(&bin, 2); w(&nc, 2); w(&flag, 2); w(&l_seq, 4); w(&mtid, 4); w(&mpos, 4); w(&tlen, 4);
    w(r.qname.data(), r.qname.size()); w(&zero, 1);
    if (n_cig) w(r.cigar.data(), 4 * n_cig);
    if (!r.seq4.empty()) w(r.seq4.data(), r.seq4.size());
    if (!r.qual.empty()) w(r.qual.data(), r.qual.size());
    if (!r.aux.empty()) w(r.aux.data(), r.aux.size());
}


}  // namespace bamout
