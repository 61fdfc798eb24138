// sortdedup_route.cpp -- host router that splits ONE record set over several GPUs
// (mgx_sortdedup_route / _routed_shard / _merge, include/mgx_sortdedup.h).
//
// What the reference's three range partitioners and its global bitmap do inside one address space
// (sortmardup/tbb/range_partitioner.h:98-100 selectPartition = partition_key / range_size;
// sortmardup/tbb/bam_partitioner.cpp:31-33; sortmardup/main.cpp:160-192):
//   * bam_partitioner      every record goes to the partition of its unified coordinate
//   * double_partitioner   every pair goes to the partition of its sort key = the smaller 5' end
//   * single_partitioner   every fragment goes to the partition of its 5' end
//   * double_pair_indicator  ONE 4L-bit map every worker sets both ends of every pair in
// becomes, for shards that do not share memory:
//   * ordering half   (coordinate, arrival index) of the records whose coordinate lies in the shard's range
//   * marking half    the templates keyed in the shard's range, as records with shard-local mate indices
//   * marks           for every pair, each 5' end that lies in ANOTHER shard's range is sent to that shard as
//                     (position, strand half): the only data that crosses shards, 8 bytes per straddling end
// Concatenating the shards' orders in shard order is the global order; duplicate flags come back per
// marking record and are scattered to arrival indices by mgx_sortdedup_merge.
#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../../include/mgx_sortdedup.h"
#include "mgx_common.h"

using mgx::set_error;

namespace {
constexpr uint16_t kIgnorable = 0x4 | 0x100 | 0x800;

struct ShardData {
    std::vector<uint64_t> order_coord;
    std::vector<uint32_t> order_arrival;
    std::vector<mgx_rec_t> mark_recs;
    std::vector<uint32_t> mark_arrival;
    std::vector<uint64_t> marks;
    uint64_t n_order = 0, n_mark = 0, n_marks = 0;   // counts are known for every shard, data only for the kept ones
    bool kept = false;
};

// per chunk and shard: how many ordering records / marking records / marks the chunk contributes
struct Counts { uint64_t order = 0, mark = 0, marks = 0; };

}  // namespace

struct mgx_sortdedup_routed {
    uint64_t L = 0, n_records = 0, width = 1;
    uint32_t n_shards = 1;
    std::vector<ShardData> shards;
    std::vector<uint64_t> order_base;      // [n_shards + 1]: where each shard's order starts in the global output
};

namespace {

inline uint32_t shard_of(const mgx_sortdedup_routed& r, uint64_t pos) {
    const uint64_t s = pos / r.width;
    return (uint32_t)(s < r.n_shards ? s : r.n_shards - 1);      // positions at or beyond L (unmapped, clipped past the end,
}                                                                 // wrapped below zero) belong to the last shard

// The two ends of a pair as the bitmap sees them (pair.cpp:71-108 + main.cpp:181-192): the smaller 5' end first,
// FF FR RF RR, RF with equal ends counts as FR.
struct PairEnds { uint64_t p1, p2; bool rev1, rev2; };
inline PairEnds pair_ends(const mgx_rec_t& a, const mgx_rec_t& b) {
    uint64_t p1 = a.prime5, p2 = b.prime5;
    bool f1 = !(a.flag & 0x10), f2 = !(b.flag & 0x10);
    if (p1 > p2) { std::swap(p1, p2); std::swap(f1, f2); }
    unsigned orient = f1 ? (f2 ? 0u : 1u) : (f2 ? 2u : 3u);
    if (p1 == p2 && orient == 2u) orient = 1u;
    return PairEnds{p1, p2, !(orient == 0u || orient == 1u), !(orient == 0u || orient == 2u)};
}

// One walk over records [a, b): `emit` decides what happens with each routed item, so that the counting pass
// and the writing pass cannot disagree.
template <class OnOrder, class OnTemplate, class OnMark>
int walk(const mgx_sortdedup_routed& R, const mgx_rec_t* recs, uint64_t a, uint64_t b, OnOrder on_order, OnTemplate on_template, OnMark on_mark) {
    for (uint64_t i = a; i < b; ++i) {
        const mgx_rec_t& r = recs[i];
        on_order(shard_of(R, r.coord), i);
        if (r.flag & kIgnorable) continue;
        if (r.mate == MGX_NO_MATE) { on_template(shard_of(R, r.prime5), i, (uint64_t)MGX_NO_MATE); continue; }
        if (r.mate >= R.n_records) return -EINVAL;
        if (r.mate < i) continue;                                  // record 2: travels with record 1
        const mgx_rec_t& m = recs[r.mate];
        const PairEnds e = pair_ends(r, m);
        const uint32_t s = shard_of(R, e.p1);
        on_template(s, i, (uint64_t)r.mate);
        const uint32_t s1 = shard_of(R, e.p1), s2 = shard_of(R, e.p2);   // s1 == s by construction
        if (s1 != s) on_mark(s1, (e.p1 << 1) | (e.rev1 ? 1u : 0u));
        if (s2 != s) on_mark(s2, (e.p2 << 1) | (e.rev2 ? 1u : 0u));
    }
    return 0;
}

}  // namespace

extern "C" {

int mgx_sortdedup_route(uint64_t L, uint64_t n_records, const mgx_rec_t* recs, uint32_t n_shards, int only_shard,
                        mgx_sortdedup_routed_t** out) {
    if (!out) { set_error("out is NULL"); return -EINVAL; }
    *out = nullptr;
    if (n_shards == 0 || n_shards > 1024) { set_error("n_shards must be in 1..1024"); return -EINVAL; }
    if (only_shard >= (int)n_shards) { set_error("only_shard %d out of range", only_shard); return -EINVAL; }
    if (n_records && !recs) { set_error("recs is NULL"); return -EINVAL; }
    if (n_records >= 0xFFFFFFF0ull) { set_error("more than 2^32 records"); return -E2BIG; }
    mgx_sortdedup_routed* R = new (std::nothrow) mgx_sortdedup_routed;
    if (!R) return -ENOMEM;
    R->L = L; R->n_records = n_records; R->n_shards = n_shards;
    R->width = (L + n_shards) / n_shards;                       // ceil((L + 1) / n_shards): coordinate L (unmapped) is in range
    if (R->width == 0) R->width = 1;
    R->shards.resize(n_shards);
    for (uint32_t k = 0; k < n_shards; ++k) R->shards[k].kept = only_shard < 0 || (int)k == only_shard;

    static const int n_thr = [] { const char* e = getenv("MGX_ROUTE_THREADS"); int v = e ? atoi(e) : (int)std::thread::hardware_concurrency(); return v < 1 ? 1 : (v > 32 ? 32 : v); }();
    const uint64_t n_chunks = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)n_thr, n_records / 65536 + 1));
    const uint64_t per = (n_records + n_chunks - 1) / n_chunks;
    std::vector<std::vector<Counts>> cnt(n_chunks, std::vector<Counts>(n_shards));
    std::vector<int> rcs(n_chunks, 0);
    auto run_chunks = [&](auto body) {
        std::vector<std::thread> th;
        for (uint64_t c = 1; c < n_chunks; ++c) th.emplace_back(body, c);
        body(0);
        for (auto& t : th) t.join();
    };
    // pass 1: counts per chunk and shard
    run_chunks([&](uint64_t c) {
        const uint64_t a = std::min(n_records, c * per), b = std::min(n_records, a + per);
        std::vector<Counts>& k = cnt[c];
        rcs[c] = walk(*R, recs, a, b,
                      [&](uint32_t s, uint64_t) { k[s].order++; },
                      [&](uint32_t s, uint64_t, uint64_t m) { k[s].mark += m == MGX_NO_MATE ? 1 : 2; },
                      [&](uint32_t s, uint64_t) { k[s].marks++; });
    });
    for (int rc : rcs) if (rc) { delete R; set_error("a record's mate index is outside the record set"); return rc; }
    // chunk-major prefix inside every shard keeps arrival order
    std::vector<std::vector<Counts>> base(n_chunks, std::vector<Counts>(n_shards));
    for (uint32_t s = 0; s < n_shards; ++s) {
        Counts run;
        for (uint64_t c = 0; c < n_chunks; ++c) {
            base[c][s] = run;
            run.order += cnt[c][s].order; run.mark += cnt[c][s].mark; run.marks += cnt[c][s].marks;
        }
        ShardData& d = R->shards[s];
        d.n_order = run.order; d.n_mark = run.mark; d.n_marks = run.marks;
        if (d.kept) {
            d.order_coord.resize(run.order); d.order_arrival.resize(run.order);
            d.mark_recs.resize(run.mark); d.mark_arrival.resize(run.mark); d.marks.resize(run.marks);
        }
    }
    R->order_base.assign(n_shards + 1, 0);
    for (uint32_t s = 0; s < n_shards; ++s) R->order_base[s + 1] = R->order_base[s] + R->shards[s].n_order;
    // pass 2: write
    run_chunks([&](uint64_t c) {
        const uint64_t a = std::min(n_records, c * per), b = std::min(n_records, a + per);
        std::vector<Counts> at = base[c];
        walk(*R, recs, a, b,
             [&](uint32_t s, uint64_t i) {
                 ShardData& d = R->shards[s];
                 const uint64_t k = at[s].order++;
                 if (d.kept) { d.order_coord[k] = recs[i].coord; d.order_arrival[k] = (uint32_t)i; }
             },
             [&](uint32_t s, uint64_t i, uint64_t m) {
                 ShardData& d = R->shards[s];
                 const uint64_t k = at[s].mark;
                 at[s].mark += m == MGX_NO_MATE ? 1 : 2;
                 if (!d.kept) return;
                 d.mark_recs[k] = recs[i]; d.mark_arrival[k] = (uint32_t)i;
                 if (m == MGX_NO_MATE) { d.mark_recs[k].mate = MGX_NO_MATE; return; }
                 d.mark_recs[k].mate = (uint32_t)(k + 1);
                 d.mark_recs[k + 1] = recs[m]; d.mark_recs[k + 1].mate = (uint32_t)k; d.mark_arrival[k + 1] = (uint32_t)m;
             },
             [&](uint32_t s, uint64_t mark) {
                 ShardData& d = R->shards[s];
                 const uint64_t k = at[s].marks++;
                 if (d.kept) d.marks[k] = mark;
             });
    });
    *out = R;
    return 0;
}

void mgx_sortdedup_routed_free(mgx_sortdedup_routed_t* r) { delete r; }

int mgx_sortdedup_routed_shard(const mgx_sortdedup_routed_t* r, uint32_t k, mgx_sortdedup_shard_t* out) {
    if (!r || !out) { set_error("NULL argument"); return -EINVAL; }
    if (k >= r->n_shards) { set_error("shard %u out of range", k); return -EINVAL; }
    const ShardData& d = r->shards[k];
    mgx_sortdedup_shard_t s{};
    s.n_order = d.n_order; s.n_mark = d.n_mark; s.n_marks = d.n_marks;
    s.order_base = r->order_base[k];
    s.coord_lo = (uint64_t)k * r->width;
    s.coord_hi = k + 1 == r->n_shards ? ~0ull : (uint64_t)(k + 1) * r->width;
    if (d.kept) {
        s.order_coord = d.order_coord.data(); s.order_arrival = d.order_arrival.data();
        s.mark_recs = d.mark_recs.data(); s.mark_arrival = d.mark_arrival.data(); s.marks = d.marks.data();
    }
    *out = s;
    return 0;
}

int mgx_sortdedup_merge(const mgx_sortdedup_routed_t* r, uint32_t k, const uint32_t* shard_order, const uint8_t* shard_dup,
                        uint32_t* out_order, uint8_t* out_dup) {
    if (!r) { set_error("routed is NULL"); return -EINVAL; }
    if (k >= r->n_shards) { set_error("shard %u out of range", k); return -EINVAL; }
    const ShardData& d = r->shards[k];
    if (!d.kept) { set_error("shard %u was not materialised by mgx_sortdedup_route", k); return -EINVAL; }
    if (out_order && shard_order && d.n_order) memcpy(out_order + r->order_base[k], shard_order, d.n_order * sizeof(uint32_t));
    if (out_dup && shard_dup)
        for (uint64_t i = 0; i < d.n_mark; ++i) if (shard_dup[i]) out_dup[d.mark_arrival[i]] = 1;
    return 0;
}

}  // extern "C"
