// sortdedup_pack.cpp -- host half of the sort/mark-duplicate path (mgx_sortdedup_pack).
//
// Turns parsed alignment records (what a SAM/BAM reader holds) into the 32-byte keys the device
// pipeline consumes, reproducing the reference's record-level semantics:
//   * ignorable records: any of UNMAP | SECONDARY | SUPPLEMENTARY  (sortmardup/tbb/bam_parser.cpp:54-58)
//   * mate discovery: the queue head is popped as record 1; record 2 is the first non-ignorable
//     record still queued whose qname equals record 1's, scanning forward while qnames are equal
//     (bam_parser.cpp:85-113); an ignorable record 1 is never paired
//   * arrival order: record 1, then record 2 if any (sortmardup/main.cpp:160-178); this is the
//     order the reference's stable coordinate sort preserves among equal coordinates when run with
//     one shuffle thread
//   * keys: BAMRecord::score / get_unify_coordinate / prime5_pos (tbb/bam_record.cpp:7-62) and the
//     Illumina tile/x/y fields of the qname (tbb/pair.cpp:11-49)
#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mgx_sortdedup.h"
#include "mgx_common.h"

extern "C" const char* mgx_last_error(void);

namespace {

constexpr uint16_t kIgnorable = 0x4 | 0x100 | 0x800;

inline uint16_t base_quality_score(const uint8_t* q, uint64_t n) {
    uint16_t s = 0;                       // wraps like the reference's uint16_t accumulator
    for (uint64_t i = 0; i < n; ++i) s = (uint16_t)(s + (q[i] >= 15 ? q[i] : 0));
    return s;
}

// which CIGAR operations consume the reference: M D N = X   (htslib BAM_CIGAR_TYPE, bit 1)
inline bool consumes_reference(uint32_t op) { return op == 0 || op == 2 || op == 3 || op == 7 || op == 8; }
inline bool is_clip(uint32_t op) { return op == 4 || op == 5; }

uint64_t unclipped_five_prime(uint64_t coord, bool forward, const uint32_t* cigar, uint64_t n_cigar) {
    if (n_cigar == 0) return coord;
    uint64_t p = coord;
    if (forward) {
        for (uint64_t i = 0; i < n_cigar && is_clip(cigar[i] & 15u); ++i) p -= cigar[i] >> 4;
        return p;
    }
    uint64_t i = n_cigar;
    while (i > 0 && is_clip(cigar[i - 1] & 15u)) { p += cigar[i - 1] >> 4; --i; }
    while (i > 0) { if (consumes_reference(cigar[i - 1] & 15u)) p += cigar[i - 1] >> 4; --i; }
    return p - 1;
}

// strtol(token, &end, 10) truncated to 16 bits, as str_to_uint16 does (pair.cpp:11-19)
inline uint16_t token_to_u16(const char* s, size_t len) {
    if (len > 0 && len <= 18) {                      // all digits, no overflow: what strtol returns, without the call
        uint64_t v = 0; size_t i = 0;
        for (; i < len; ++i) { const unsigned d = (unsigned)(s[i] - '0'); if (d > 9) break; v = v * 10 + d; }
        if (i == len) return (uint16_t)v;
    }
    char buf[64];
    if (len >= sizeof buf) len = sizeof buf - 1;
    memcpy(buf, s, len);
    buf[len] = 0;
    return (uint16_t)strtol(buf, nullptr, 10);
}

// strtok_r(":") semantics: empty fields between consecutive ':' do not count as tokens
void tile_x_y(const char* q, uint64_t len, uint16_t out[3]) {
    const char* tok[8]; size_t tl[8];
    int n = 0;
    uint64_t i = 0;
    while (i < len) {
        while (i < len && q[i] == ':') ++i;
        if (i >= len) break;
        uint64_t b = i;
        while (i < len && q[i] != ':') ++i;
        if (n < 8) { tok[n] = q + b; tl[n] = i - b; }
        ++n;
    }
    out[0] = out[1] = out[2] = 0;
    const int first = n == 7 ? 4 : (n == 6 ? 3 : -1);
    if (first < 0) return;
    for (int k = 0; k < 3; ++k) out[k] = token_to_u16(tok[first + k], tl[first + k]);
}

}  // namespace

extern "C" int mgx_sortdedup_pack(const mgx_raw_records_t* raw, mgx_rec_t* out, uint32_t* out_input_index,
                                  uint64_t* out_L) {
    return mgx_sortdedup_pack_scored(raw, nullptr, out, out_input_index, out_L);
}

extern "C" int mgx_sortdedup_pack_scored(const mgx_raw_records_t* raw, const uint16_t* score, mgx_rec_t* out, uint32_t* out_input_index,
                                         uint64_t* out_L) {
    if (!raw || !out_L || (raw->n_records && (!out || !out_input_index))) { mgx::set_error("NULL argument"); return -EINVAL; }
    const uint64_t n = raw->n_records;
    if (n >= 0xFFFFFFFFull) { mgx::set_error("more than 2^32-1 records"); return -E2BIG; }
    if (n && (!raw->flag || !raw->tid || !raw->pos || !raw->cigar_off || (!score && !raw->qual_off) || !raw->qname_off ||
              (raw->n_targets && !raw->target_len))) { mgx::set_error("NULL array in raw records"); return -EINVAL; }
    // offsets index host memory: a decreasing table would turn into an out-of-bounds read below
    for (uint64_t r = 0; r < n; ++r)
        if (raw->cigar_off[r + 1] < raw->cigar_off[r] || (!score && raw->qual_off[r + 1] < raw->qual_off[r]) || raw->qname_off[r + 1] < raw->qname_off[r]) {
            mgx::set_error("record %llu: offset table is not monotonic", (unsigned long long)r);
            return -EINVAL;
        }
    std::vector<uint64_t> ktable(raw->n_targets + 1);
    uint64_t acc = 0;
    for (uint32_t t = 0; t < raw->n_targets; ++t) { ktable[t] = acc; acc += raw->target_len[t]; }
    ktable[raw->n_targets] = acc;
    *out_L = acc;

    auto qlen = [&](uint64_t i) { return raw->qname_off[i + 1] - raw->qname_off[i]; };
    auto same_qname = [&](uint64_t a, uint64_t b) {
        return qlen(a) == qlen(b) && memcmp(raw->qname + raw->qname_off[a], raw->qname + raw->qname_off[b], qlen(a)) == 0;
    };
    auto emit = [&](uint64_t k, uint64_t r, uint32_t mate) -> int {
        const int32_t tid = raw->tid[r];
        if (tid >= (int32_t)raw->n_targets) { mgx::set_error("record %llu: tid %d out of range", (unsigned long long)r, tid); return -EINVAL; }
        mgx_rec_t& o = out[k];
        memset(&o, 0, sizeof o);
        o.coord = tid < 0 ? acc : ktable[tid] + (uint64_t)raw->pos[r];
        o.flag = raw->flag[r];
        o.prime5 = unclipped_five_prime(o.coord, (o.flag & 0x10) == 0, raw->cigar + raw->cigar_off[r],
                                        raw->cigar_off[r + 1] - raw->cigar_off[r]);
        o.score = score ? score[r] : base_quality_score(raw->qual + raw->qual_off[r], raw->qual_off[r + 1] - raw->qual_off[r]);
        uint16_t t[3];
        tile_x_y(raw->qname + raw->qname_off[r], qlen(r), t);
        o.tile = t[0]; o.x = t[1]; o.y = t[2];
        o.mate = mate;
        out_input_index[k] = (uint32_t)r;
        return 0;
    };

    // The queue logic only ever looks ahead inside one run of equal qnames, so the input can be cut
    // at qname changes and the pieces packed independently by several threads: the number of records
    // a piece emits equals the number it consumes, hence arrival slot == input slot of the piece's
    // first record, and mate indices are piece-local offsets plus that base.
    auto pack_range = [&](uint64_t lo, uint64_t hi) -> int {
        std::vector<uint8_t> taken(hi - lo, 0);
        uint64_t k = lo;
        for (uint64_t i = lo; i < hi; ++i) {
            if (taken[i - lo]) continue;
            taken[i - lo] = 1;
            uint64_t mate = hi;
            if (!(raw->flag[i] & kIgnorable)) {
                for (uint64_t q = i + 1; q < hi && (taken[q - lo] || same_qname(i, q)); ++q) {
                    if (taken[q - lo]) continue;
                    if (!(raw->flag[q] & kIgnorable)) { mate = q; break; }
                }
            }
            int rc;
            if (mate == hi) {
                if ((rc = emit(k, i, MGX_NO_MATE))) return rc;
                k += 1;
            } else {
                taken[mate - lo] = 1;
                if ((rc = emit(k, i, (uint32_t)(k + 1)))) return rc;
                if ((rc = emit(k + 1, mate, (uint32_t)k))) return rc;
                k += 2;
            }
        }
        return 0;
    };
    unsigned T = std::thread::hardware_concurrency();
    if (const char* e = getenv("MGX_PACK_THREADS")) T = (unsigned)atoi(e);
    if (T < 1) T = 1;
    if (T > 64) T = 64;
    if (n < 200000 || T == 1) return pack_range(0, n);
    std::vector<uint64_t> cut(T + 1, n);
    cut[0] = 0;
    for (unsigned t = 1; t < T; ++t) {
        uint64_t p = n * t / T;
        while (p < n && p > 0 && same_qname(p - 1, p)) ++p;       // never split a qname group
        cut[t] = std::max(p, cut[t - 1]);
    }
    std::vector<int> rcs(T, 0);
    std::vector<std::string> errs(T);
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < T; ++t)
        pool.emplace_back([&, t]() { rcs[t] = pack_range(cut[t], cut[t + 1]); if (rcs[t]) errs[t] = mgx_last_error(); });
    for (auto& th : pool) th.join();
    for (unsigned t = 0; t < T; ++t) if (rcs[t]) { mgx::set_error("%s", errs[t].c_str()); return rcs[t]; }
    return 0;
}
