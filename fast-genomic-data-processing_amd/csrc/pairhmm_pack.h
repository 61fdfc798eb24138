// pairhmm_pack.h -- host packer of the PairHMM work queue: cuts test cases [lo, hi) out of a long
// stream into a self-contained batch.
//
// The reference's worker threads each pull the next active region off an atomic index and hand its
// test cases to the native layer (deepmutect/Mutect2Cpp-master/src/main.cpp:254, 302-315;
// utils/pairhmm/VectorLoglessPairHMM.cpp:88-119).  Here a worker pulls the next BATCH of test cases:
// the reads and haplotypes those test cases reference are gathered once each, in first-use order,
// into contiguous arrays (the only bytes that cross PCIe), and the test cases are re-expressed in
// local indices.  Pure host code: no HIP, no device.
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

#include <cerrno>

#include "../../include/mgx_pairhmm.h"
#include "mgx_common.h"

namespace mgx {

// argument checks shared by every entry point that takes an mgx_pairhmm_input_t
inline int validate(const mgx_pairhmm_input_t* in) {
    if (!in) { set_error("input is NULL"); return -EINVAL; }
    if (in->n_pairs == 0 && (in->pair_read || in->n_reads == 0 || in->n_haps == 0)) return 0;
    if (!in->read_off || !in->hap_off || !in->bases || !in->qual || !in->ins || !in->del ||
        !in->gcp || !in->hap_bases || (!in->pair_read != !in->pair_hap)) {
        set_error("a required input array is NULL");
        return -EINVAL;
    }
    if (!in->pair_read && (in->n_reads == 0 || in->n_haps == 0)) {
        set_error("%llu test cases without pair arrays and without reads or haplotypes", (unsigned long long)in->n_pairs);
        return -EINVAL;
    }
    if (in->n_pairs > 0xFFFFFFF0ull) { set_error("more than 2^32 test cases in one batch"); return -E2BIG; }
    // offsets index device memory: a decreasing table would turn into an out-of-bounds access there
    for (uint64_t r = 0; r < in->n_reads; ++r)
        if (in->read_off[r + 1] < in->read_off[r]) { set_error("read_off is not monotonic at %llu", (unsigned long long)r); return -EINVAL; }
    for (uint64_t h = 0; h < in->n_haps; ++h)
        if (in->hap_off[h + 1] < in->hap_off[h]) { set_error("hap_off is not monotonic at %llu", (unsigned long long)h); return -EINVAL; }
    return 0;
}


struct PackPlan {
    uint64_t lo = 0, hi = 0;
    std::vector<uint64_t> lread, lhap;           // local index -> index in the caller's input
    std::vector<uint64_t> roff{0}, hoff{0};      // local prefix offsets (bytes) of the gathered arrays
    std::vector<uint32_t> pair_read, pair_hap;   // local indices, one per test case of [lo, hi)
};

namespace detail {
// u64 -> u32 open-addressing map sized for one batch
struct IndexMap {
    std::vector<uint64_t> key;    // stored + 1 (0 = empty)
    std::vector<uint32_t> val;
    uint64_t mask = 0;
    void reset(uint64_t n_items) {
        uint64_t cap = 64;
        while (cap < 2 * n_items) cap <<= 1;
        key.assign(cap, 0); val.resize(cap); mask = cap - 1;
    }
    // returns the slot's value, inserting `fresh` when the key is new (*is_new set)
    uint32_t get_or_put(uint64_t k, uint32_t fresh, bool* is_new) {
        uint64_t h = (k * 0x9E3779B97F4A7C15ull) >> 20;
        for (;; ++h) {
            const uint64_t s = h & mask;
            if (key[s] == k + 1) { *is_new = false; return val[s]; }
            if (key[s] == 0) { key[s] = k + 1; val[s] = fresh; *is_new = true; return fresh; }
        }
    }
};
}  // namespace detail

// Test case i of `in`: (read, haplotype) indices; the cross-product form enumerates read-major.
inline void pack_pair_of(const mgx_pairhmm_input_t* in, uint64_t i, uint64_t* r, uint64_t* h) {
    if (in->pair_read) { *r = in->pair_read[i]; *h = in->pair_hap[i]; }
    else { *r = i / in->n_haps; *h = i % in->n_haps; }
}
inline uint64_t pack_n_pairs(const mgx_pairhmm_input_t* in) {
    return in->pair_read ? in->n_pairs : in->n_reads * in->n_haps;
}

// Returns 0, or the (1-based) position of the first test case whose indices are out of range.
inline uint64_t pack_plan(const mgx_pairhmm_input_t* in, uint64_t lo, uint64_t hi, PackPlan* p) {
    const uint64_t n = hi - lo;
    p->lo = lo; p->hi = hi;
    p->lread.clear(); p->lhap.clear(); p->roff.assign(1, 0); p->hoff.assign(1, 0);
    p->pair_read.resize(n); p->pair_hap.resize(n);
    detail::IndexMap rmap, hmap;
    rmap.reset(n); hmap.reset(n);
    uint64_t last_r = ~0ull, last_h = ~0ull;
    uint32_t last_lr = 0, last_lh = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t r, h;
        pack_pair_of(in, lo + i, &r, &h);
        if (r >= in->n_reads || h >= in->n_haps) return i + 1;
        if (r != last_r) {                          // consecutive test cases of one read: no lookup
            bool is_new;
            last_lr = rmap.get_or_put(r, (uint32_t)p->lread.size(), &is_new);
            if (is_new) { p->lread.push_back(r); p->roff.push_back(p->roff.back() + (in->read_off[r + 1] - in->read_off[r])); }
            last_r = r;
        }
        if (h != last_h) {
            bool is_new;
            last_lh = hmap.get_or_put(h, (uint32_t)p->lhap.size(), &is_new);
            if (is_new) { p->lhap.push_back(h); p->hoff.push_back(p->hoff.back() + (in->hap_off[h + 1] - in->hap_off[h])); }
            last_h = h;
        }
        p->pair_read[i] = last_lr; p->pair_hap[i] = last_lh;
    }
    return 0;
}

// Gathers the planned reads / haplotypes into the destination arrays (sized roff.back() / hoff.back()).
// Runs of sequences that are neighbours in the source are copied with one memcpy per array.
inline void pack_copy(const mgx_pairhmm_input_t* in, const PackPlan& p, uint8_t* bases, uint8_t* qual, uint8_t* ins, uint8_t* del,
                      uint8_t* gcp, uint8_t* hap) {
    const size_t nr = p.lread.size(), nh = p.lhap.size();
    for (size_t a = 0; a < nr;) {
        size_t b = a + 1;
        while (b < nr && p.lread[b] == p.lread[b - 1] + 1) ++b;
        const uint64_t src = in->read_off[p.lread[a]], len = in->read_off[p.lread[b - 1] + 1] - src, dst = p.roff[a];
        memcpy(bases + dst, in->bases + src, len); memcpy(qual + dst, in->qual + src, len);
        memcpy(ins + dst, in->ins + src, len);     memcpy(del + dst, in->del + src, len);
        memcpy(gcp + dst, in->gcp + src, len);
        a = b;
    }
    for (size_t a = 0; a < nh;) {
        size_t b = a + 1;
        while (b < nh && p.lhap[b] == p.lhap[b - 1] + 1) ++b;
        const uint64_t src = in->hap_off[p.lhap[a]], len = in->hap_off[p.lhap[b - 1] + 1] - src;
        memcpy(hap + p.hoff[a], in->hap_bases + src, len);
        a = b;
    }
}

}  // namespace mgx
