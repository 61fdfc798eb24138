// mgx_sortdedup.hip -- device half of the sort / mark-duplicate path for MI355X (gfx950).
//
// Re-expresses sortmardup/main.cpp:235-388 (three per-partition std::sort/std::stable_sort calls,
// two run scans, two bitmaps) as an HBM-bound pipeline over packed 32-byte records:
//
//   build      records -> coordinate keys, compacted double-pair / single-pair entries
//              (pair.cpp:51-108 key rules) + indicator bits (main.cpp:181-192) + key maxima
//   radix      stable LSD radix sort, 8-bit digits, 4096-key tiles: per-tile histogram ->
//              column-wise scan -> ranked scatter through LDS (used for: doubles by mate 5' end,
//              doubles by sort_key, singles by sort_key, records by unified coordinate)
//   best-of-run  every run of equal keys keeps its best entry (score desc, tile/x/y asc, arrival
//              asc), the rest are duplicates (main.cpp:269-280, 319-340); singles additionally
//              test the indicator bitmap
//
// No collective and no host round trip inside the pipeline except one 64-byte read-back of the
// entry counts and key maxima (they size the sorts and pick the number of digit passes).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <new>
#include <thread>
#include <vector>

#include "../../include/mgx_sortdedup.h"
#include "mgx_common.h"

using mgx::set_error;

namespace {

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,     \
                      __LINE__);                                                           \
            return -EIO;                                                                   \
        }                                                                                  \
    } while (0)

using u64 = unsigned long long;   // == uint64_t on this ABI; the type HIP atomics are declared for
using u32 = uint32_t;

#ifndef MGX_SCATTER_WAVES
#define MGX_SCATTER_WAVES 4
#endif
constexpr int kScatterWaves = MGX_SCATTER_WAVES;   // wavefronts per scatter workgroup
constexpr int kTileThreads = 256;
constexpr int kItems = 16;
constexpr int kTile = kTileThreads * kItems;          // 4096 keys per workgroup
constexpr int kChunkTiles = 128;                      // tiles per column-scan chunk
constexpr u32 kIgnorable = 0x4 | 0x100 | 0x800;
constexpr int kWalkCap = 64;                          // longest run a single lane walks

// Device-side scalars, one small read-back.  Every workgroup of the build kernel bumps the entry
// counters, so each one sits in its own 128-byte line (atomics on one line serialise at ~90/us and
// would otherwise also stall the plain reads of the maxima next to them).
struct Scalars {
    u64 max_coord, max_k1d, max_k2d, max_k1s, max_near;
    u32 n_long_d, n_long_s, n_long_n, n_dup;
    u32 bad_mate;                              // a record names a mate index outside the shard
    u32 pad0_[17];
    alignas(128) u32 n_double; u32 pad1_[31];  // "far" double pairs (two-word keys)
    alignas(128) u32 n_single; u32 pad2_[31];
    alignas(128) u32 n_near;   u32 pad3_[31];
    alignas(128) u32 n_multi_d; u32 n_multi_s, n_multi_n;                   // run heads with more than one entry
    u32 huge_runs; u32 pad4_[28];            // a position holds more near pairs than the in-run comparison accepts
};
// near pair: mate 5' end less than kNearSpan beyond record 1's (every proper pair; insert sizes are a
// few hundred bases).  14 delta bits + 2 orientation bits + a 32-bit position = 48 key bits: six
// 8-bit passes (a 16-bit delta would cost a seventh).
constexpr int kNearDeltaBits = 14;
constexpr u64 kNearSpan = 1ull << kNearDeltaBits;
// The identity occupies the upper 48 bits of the key word and only those are sorted; the 16 bits
// below ride along for free and carry 0xFFFF - (pair score), so the duplicate search knows the better
// pair of a run from the sorted keys alone and gathers records only for the losers' mate index (and
// on the rare score tie).
constexpr int kNearScoreBits = 16;
constexpr int kNearDeltaShift = kNearScoreBits;                       // delta
constexpr int kNearOrientShift = kNearScoreBits + kNearDeltaBits;     // orientation
constexpr int kNearShift = kNearOrientShift + 2;                      // record 1's 5' end
// Near pairs are SORTED on record 1's 5' end alone (the upper 32 bits: four passes instead of six).  All
// pairs that start at one position form a run; inside it the duplicate search groups by the remaining
// identity bits (orientation, insert) by comparing every entry with every other one -- runs are two or
// three entries, and a position with more than kHugeRun pairs (amplicon-like data) makes the whole
// pipeline fall back to the full six-pass sort, where equal identities are adjacent.
constexpr u32 kHugeRun = 4096;

__device__ __forceinline__ u64 lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// exclusive scan of one u32 per thread over a 256-thread block (4 wavefronts); returns the
// exclusive prefix, *total gets the block sum.  sm must hold 4 u32.
__device__ __forceinline__ u32 block_excl_scan_256(u32 v, u32* sm, u32* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        u32 t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    __syncthreads();
    if (lane == 63) sm[wave] = incl;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const u32 s = sm[w]; if (w < wave) base += s; tot += s; }
    *total = tot;
    return base + incl - v;
}

// ---------------------------------------------------------------------------------------------
// build: classify records, emit pair entries and coordinate keys
// ---------------------------------------------------------------------------------------------
constexpr int kBuildBlock = 2048;

struct BuildOut {
    u64* ckey; u32* cval;                 // coordinate sort input
    u64* dk1; u64* dk2; u32* drec;        // double-pair entries (compacted, arrival order)
    u64* sk1; u32* srec;                  // single-pair entries
    u32* indicator; u64 indicator_bits;   // double_pair_indicator, 4L bits
    u64 L;
    int packed_coord;                     // ckey = coord << 32 | i (every coordinate < 2^32)
    int packed_pair;                      // dk2 = mate 5' end << 32 | record (every 5' end < 2^32)
    u64* nk; u32* nrec;                   // near double pairs: one key word p1 << 32 | orient << 30 | (p2 - p1) << 16 | inverted pair score
    int near_enabled;
    u32 mate_flag;                        // 0x80000000 when record indices leave bit 31 free: "the mate is the neighbour rec ^ 1"
    u32* half_hist;                       // [workgroups][256]: histogram of the coordinate's low byte (first digit of the record sort), or NULL
};

// Entries are compacted with one global atomic per workgroup and kind, so their order is not the
// arrival order; nothing downstream depends on it: runs are formed by key equality and total ties
// between entries are broken by the record index itself (k_mark_list), not by position.
//
// Single pass over the records: a workgroup reads its 2048 records once (all loads issued up
// front), classifies them, keeps the derived entry words in registers while the per-kind counts
// are settled (LDS atomics give block-local slots, ONE global atomic per kind gives the block's
// base -- a single counter word sustains only ~90 atomics/us, never one per wavefront), then
// stores the entries.  The mate's record is read through the cache: mates are adjacent in arrival
// order, so the line is already there.
__global__ __launch_bounds__(256) void k_build_emit(const mgx_rec_t* __restrict__ recs, u32 n, BuildOut o, Scalars* sc) {
    constexpr int ITEMS = kBuildBlock / 256;
    __shared__ u64 smax[4][5];
    __shared__ u32 s_cnt[3], s_base[3];
    __shared__ u32 s_hist[256];
    const u32 base = blockIdx.x * kBuildBlock;
    const u64 lt = lanemask_lt();
    if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
    s_hist[threadIdx.x] = 0;
    __syncthreads();

    u64 coord[ITEMS], p5[ITEMS];
    u32 mate[ITEMS], flag[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 i = base + k * 256 + threadIdx.x;
        coord[k] = 0; p5[k] = 0; mate[k] = MGX_NO_MATE; flag[k] = kIgnorable;      // flag: SAM flag | score << 16
        if (i < n) { coord[k] = recs[i].coord; p5[k] = recs[i].prime5; mate[k] = recs[i].mate; flag[k] = (u32)recs[i].flag | (u32)recs[i].score << 16; }
    }
    u64 mp5[ITEMS]; u32 mflag[ITEMS];      // mflag: mate's flag | mate's score << 16
    bool bad = false;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 i = base + k * 256 + threadIdx.x;
        if (i < n && mate[k] != MGX_NO_MATE && mate[k] >= n) { bad = true; mate[k] = MGX_NO_MATE; flag[k] |= kIgnorable; }
        mp5[k] = 0; mflag[k] = 0;
        // only record 1 of a pair (mate > i, not ignorable) looks at its mate
        if (i < n && !(flag[k] & kIgnorable) && mate[k] != MGX_NO_MATE && mate[k] > i) { mp5[k] = recs[mate[k]].prime5; mflag[k] = (u32)recs[mate[k]].flag | (u32)recs[mate[k]].score << 16; }
    }
    if (bad) sc->bad_mate = 1;

    u64 ka[ITEMS], kb[ITEMS];       // entry words: near key | sort_key ; mate end (far doubles only)
    u32 slot[ITEMS];                // block-local slot within the entry's kind
    int cls[ITEMS];                 // 0 none, 1 far double, 2 single, 3 near double
    u64 m_coord = 0, m_k1d = 0, m_k2d = 0, m_k1s = 0, m_near = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 i = base + k * 256 + threadIdx.x;
        int c = 0;
        u64 wa = 0, wb = 0;
        if (i < n) {
            if (o.ckey) {      // (a shard's ordering half is a different record subset: k_order_keys writes the keys then)
                o.ckey[i] = o.packed_coord ? (coord[k] << 32) | i : coord[k];
                if (!o.packed_coord) o.cval[i] = i;
                m_coord = max(m_coord, coord[k]);
                // the record sort's first digit is the coordinate's low byte in either key form: its per-tile
                // histogram is gathered here, where the keys are made, instead of by a pass that re-reads them
                if (o.half_hist) atomicAdd(&s_hist[(u32)coord[k] & 255u], 1u);
            }
            if (!(flag[k] & kIgnorable)) c = mate[k] == MGX_NO_MATE ? 2 : (mate[k] > i ? 1 : 0);
            if (c == 1) {
                // DoublePair::DoublePair, pair.cpp:71-108
                u64 p1 = p5[k], p2 = mp5[k];
                bool f1 = !(flag[k] & 0x10), f2 = !(mflag[k] & 0x10);
                if (p1 > p2) { const u64 t = p1; p1 = p2; p2 = t; const bool tf = f1; f1 = f2; f2 = tf; }
                u32 orient = f1 ? (f2 ? 0u : 1u) : (f2 ? 2u : 3u);       // FF FR RF RR
                if (p1 == p2 && orient == 2u) orient = 1u;
                const u64 k1 = (p1 << 2) + orient;
                m_k1d = max(m_k1d, k1); m_k2d = max(m_k2d, p2);
                // near pair: the whole (sort_key, mate end) identity fits one word, injectively
                const bool near = o.near_enabled && p2 - p1 < kNearSpan;
                const u32 pair_score = ((flag[k] >> 16) + (mflag[k] >> 16)) & 0xFFFFu;              // pair.cpp:81: uint16 sum
                const u64 nkey = (p1 << kNearShift) | ((u64)orient << kNearOrientShift) | ((p2 - p1) << kNearDeltaShift) |
                                 (u64)(0xFFFFu - pair_score);
                c = near ? 3 : 1;
                wa = near ? nkey : k1;
                wb = p2;
                if (near) m_near = max(m_near, nkey);
            } else if (c == 2) {
                // SinglePair::SinglePair, pair.cpp:51-69
                wa = (p5[k] << 2) + ((flag[k] & 0x10) ? 3u : 0u);
                m_k1s = max(m_k1s, wa);
            }
        }
        ka[k] = wa; kb[k] = wb;
        cls[k] = c;
        const u64 bd = __ballot(c == 1), bs = __ballot(c == 2), bn = __ballot(c == 3);
        u32 based = 0, bases = 0, basen = 0;
        if ((threadIdx.x & 63) == 0) {
            if (bd) based = atomicAdd(&s_cnt[0], (u32)__popcll(bd));      // LDS
            if (bs) bases = atomicAdd(&s_cnt[1], (u32)__popcll(bs));
            if (bn) basen = atomicAdd(&s_cnt[2], (u32)__popcll(bn));
        }
        based = __shfl(based, 0, 64); bases = __shfl(bases, 0, 64); basen = __shfl(basen, 0, 64);
        slot[k] = c == 1 ? based + (u32)__popcll(bd & lt) : c == 2 ? bases + (u32)__popcll(bs & lt) : basen + (u32)__popcll(bn & lt);
    }
    __syncthreads();
    if (o.half_hist) o.half_hist[(size_t)blockIdx.x * 256 + threadIdx.x] = s_hist[threadIdx.x];
    if (threadIdx.x < 3) {
        const u32 cnt = s_cnt[threadIdx.x];
        u32* dst = threadIdx.x == 0 ? &sc->n_double : threadIdx.x == 1 ? &sc->n_single : &sc->n_near;
        s_base[threadIdx.x] = cnt ? atomicAdd(dst, cnt) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 i = base + k * 256 + threadIdx.x;
        if (cls[k] == 3) {
            const u32 at = s_base[2] + slot[k];
            o.nk[at] = ka[k]; o.nrec[at] = i | (mate[k] == (i ^ 1u) ? o.mate_flag : 0u);
        } else if (cls[k] == 1) {
            const u32 at = s_base[0] + slot[k];
            o.dk1[at] = ka[k];
            if (o.packed_pair) o.dk2[at] = (kb[k] << 32) | i; else { o.dk2[at] = kb[k]; o.drec[at] = i; }
        } else if (cls[k] == 2) {
            const u32 at = s_base[1] + slot[k];
            o.sk1[at] = ka[k]; o.srec[at] = i;
        }
    }
    // block max -> global atomics, skipped when the (possibly stale) global value already covers it
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m_coord = max(m_coord, (u64)__shfl_xor(m_coord, off, 64));
        m_k1d = max(m_k1d, (u64)__shfl_xor(m_k1d, off, 64));
        m_k2d = max(m_k2d, (u64)__shfl_xor(m_k2d, off, 64));
        m_k1s = max(m_k1s, (u64)__shfl_xor(m_k1s, off, 64));
        m_near = max(m_near, (u64)__shfl_xor(m_near, off, 64));
    }
    if (lane == 0) { smax[wave][0] = m_coord; smax[wave][1] = m_k1d; smax[wave][2] = m_k2d; smax[wave][3] = m_k1s; smax[wave][4] = m_near; }
    __syncthreads();
    if (threadIdx.x < 5) {
        const u64 v = max(max(smax[0][threadIdx.x], smax[1][threadIdx.x]), max(smax[2][threadIdx.x], smax[3][threadIdx.x]));
        u64* dst = threadIdx.x == 0 ? &sc->max_coord : threadIdx.x == 1 ? &sc->max_k1d : threadIdx.x == 2 ? &sc->max_k2d
                 : threadIdx.x == 3 ? &sc->max_k1s : &sc->max_near;
        if (v > __atomic_load_n(dst, __ATOMIC_RELAXED)) atomicMax(dst, v);
    }
}

// Upload wire format: whenever a piece's coordinates and 5' ends all fit 32 bits (always, unless a 5' end wrapped
// below zero or L >= 2^32) the staging threads squeeze the 32-byte records to 24 bytes on their way into the pinned
// buffer -- a quarter fewer bytes over PCIe, the link being what an upload waits for -- and this kernel restores
// the 32-byte layout in HBM right behind the copy.
struct Wire24 { u32 coord, prime5, mate; uint16_t flag, score, tile, x, y, pad_; };
static_assert(sizeof(Wire24) == 24, "wire record is 24 bytes");
__global__ __launch_bounds__(256) void k_expand24(const Wire24* __restrict__ src, mgx_rec_t* __restrict__ dst, u32 n) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Wire24 w = src[i];
    mgx_rec_t r;
    r.coord = w.coord; r.prime5 = w.prime5; r.mate = w.mate; r.flag = w.flag; r.score = w.score;
    r.tile = w.tile; r.x = w.x; r.y = w.y; r.pad_ = 0;
    dst[i] = r;
}

// Sharded input (mgx_sortdedup_upload_shard): the records a shard ORDERS are not the records it MARKS, so the
// coordinate keys come from the routed (coordinate, global arrival index) arrays instead of the build kernel.
__global__ __launch_bounds__(256) void k_order_keys(const u64* __restrict__ coord, const u32* __restrict__ arrival, u32 n,
                                                    u64* __restrict__ ckey, u32* __restrict__ cval, int packed, Scalars* sc) {
    __shared__ u64 smax[4];
    u64 m = 0;
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const u64 cd = coord[i];
        const u32 a = arrival[i];
        ckey[i] = packed ? (cd << 32) | a : cd;
        if (!packed) cval[i] = a;
        m = max(m, cd);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (u64)__shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u64 v = max(max(smax[0], smax[1]), max(smax[2], smax[3]));
        if (v > __atomic_load_n(&sc->max_coord, __ATOMIC_RELAXED)) atomicMax(&sc->max_coord, v);
    }
}

// Indicator marks routed from other shards (5' position << 1 | reverse half): ORed into the bitmap after the
// shard's own pairs have defined it and before its fragments are tested.  In the tiled layout a position at or
// beyond L - 64 cannot be hit by any of this shard's fragments (the layout is only chosen when all of its own
// 5' ends are below), so such a mark is dropped instead of aliasing into the other half.
__global__ __launch_bounds__(256) void k_or_marks(const u64* __restrict__ marks, u32 n, u32* __restrict__ indicator, u64 ind_bits,
                                                  u64 ind_off, u64 fwd_limit) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u64 mk = marks[i], pos = mk >> 1;
    if (pos >= fwd_limit) return;
    const u64 b = pos + ((mk & 1) ? ind_off : 0ull);
    if (b < ind_bits) atomicOr(&indicator[b >> 5], 1u << (b & 31));
}

// ---------------------------------------------------------------------------------------------
// radix sort pass: histogram -> column scan -> ranked scatter
// ---------------------------------------------------------------------------------------------
template <int TILE>
__global__ __launch_bounds__(256) void k_radix_hist(const u64* __restrict__ keys, u32 n, int shift, u32* __restrict__ hist) {
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const u32 base = blockIdx.x * TILE;
#pragma unroll
    for (int k = 0; k < TILE / 256; ++k) {
        const u32 i = base + k * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(u32)(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];     // tile-major: coalesced
}

// "Onesweep" (round 3): the per-pass tile histograms and their three scan kernels are gone.  ONE pass over
// a sort's keys counts every digit of every pass (global digit totals do not depend on the order of the keys), and each
// scatter pass finds a tile's offsets by decoupled look-back over the tiles before it (k_radix_scatter<..., ONESWEEP>).
constexpr int kMaxPasses = 8;
constexpr int kTotalsBlocks = 2048;
__global__ __launch_bounds__(256) void k_radix_totals(const u64* __restrict__ keys, u32 n, int first_shift, int n_passes, u32* __restrict__ totals) {
    __shared__ u32 h[kMaxPasses * 256];
    for (int i = threadIdx.x; i < n_passes * 256; i += 256) h[i] = 0;
    __syncthreads();
    const u64 per = ((u64)n + gridDim.x - 1) / gridDim.x;
    const u64 lo = (u64)blockIdx.x * per, hi = min((u64)n, lo + per);
    for (u64 i = lo + threadIdx.x; i < hi; i += 256) {
        const u64 k = keys[i] >> first_shift;
        for (int p = 0; p < n_passes; ++p) atomicAdd(&h[p * 256 + ((u32)(k >> (8 * p)) & 255u)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_passes * 256; i += 256) if (h[i]) atomicAdd(&totals[i], h[i]);
}

// first pass of the record sort: the build kernel's half-tile histograms (kBuildBlock records each) -> tile histograms
template <int TILE>
__global__ __launch_bounds__(256) void k_hist_merge(const u32* __restrict__ half, u32 n_half, u32* __restrict__ hist) {
    constexpr int RATIO = TILE / kBuildBlock;
    u32 v = 0;
#pragma unroll
    for (int k = 0; k < RATIO; ++k) {
        const u32 hb = blockIdx.x * RATIO + k;
        if (hb < n_half) v += half[(size_t)hb * 256 + threadIdx.x];
    }
    hist[(size_t)blockIdx.x * 256 + threadIdx.x] = v;
}

// column sums per chunk of kChunkTiles tiles
__global__ __launch_bounds__(256) void k_radix_chunk_sums(const u32* __restrict__ hist, u32 n_tiles, u32* __restrict__ chunk_sums) {
    const u32 t0 = blockIdx.x * kChunkTiles, t1 = min(n_tiles, t0 + kChunkTiles);
    u32 s = 0;
    for (u32 t = t0; t < t1; ++t) s += hist[(size_t)t * 256 + threadIdx.x];
    chunk_sums[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// one block: digit bases (exclusive over digits) + exclusive scan over chunks per digit column.
// A single workgroup alone on the device is pure latency, so both loops keep 16 independent loads in
// flight per thread (the in-place update would otherwise serialise on load -> store -> load).
__global__ __launch_bounds__(256) void k_radix_scan_chunks(u32* chunk_sums, u32 n_chunks) {
    __shared__ u32 sm[4];
    constexpr int B = 16;
    u32 tot = 0;
    for (u32 c0 = 0; c0 < n_chunks; c0 += B) {
        u32 x[B];
#pragma unroll
        for (int k = 0; k < B; ++k) x[k] = c0 + k < n_chunks ? chunk_sums[(size_t)(c0 + k) * 256 + threadIdx.x] : 0u;
#pragma unroll
        for (int k = 0; k < B; ++k) tot += x[k];
    }
    u32 all;
    u32 run = block_excl_scan_256(tot, sm, &all);      // where this digit starts in the output
    for (u32 c0 = 0; c0 < n_chunks; c0 += B) {
        u32 x[B];
#pragma unroll
        for (int k = 0; k < B; ++k) x[k] = c0 + k < n_chunks ? chunk_sums[(size_t)(c0 + k) * 256 + threadIdx.x] : 0u;
#pragma unroll
        for (int k = 0; k < B; ++k) {
            if (c0 + k < n_chunks) chunk_sums[(size_t)(c0 + k) * 256 + threadIdx.x] = run;
            run += x[k];
        }
    }
}

// per chunk: running prefix down the tiles of the chunk, in place: hist[tile][d] -> global offset
__global__ __launch_bounds__(256) void k_radix_apply(u32* __restrict__ hist, u32 n_tiles, const u32* __restrict__ chunk_sums) {
    const u32 t0 = blockIdx.x * kChunkTiles, t1 = min(n_tiles, t0 + kChunkTiles);
    u32 run = chunk_sums[(size_t)blockIdx.x * 256 + threadIdx.x];
    constexpr int B = 16;
    for (u32 tb = t0; tb < t1; tb += B) {
        u32 x[B];
#pragma unroll
        for (int k = 0; k < B; ++k) x[k] = tb + k < t1 ? hist[(size_t)(tb + k) * 256 + threadIdx.x] : 0u;
#pragma unroll
        for (int k = 0; k < B; ++k) {
            if (tb + k < t1) hist[(size_t)(tb + k) * 256 + threadIdx.x] = run;
            run += x[k];
        }
    }
}

// block-wide exclusive scan over NW wavefronts (every thread of the block must call it)
template <int NW>
__device__ __forceinline__ u32 block_excl_scan(u32 v, u32* sm) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        u32 t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    __syncthreads();
    if (lane == 63) sm[wave] = incl;
    __syncthreads();
    u32 base = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const u32 s = sm[w]; if (w < wave) base += s; }
    return base + incl - v;
}

// One tile of WAVES x 1024 keys per workgroup of WAVES wavefronts; every wavefront ranks a contiguous slice.
// (8 wavefronts = 8192-key tiles double the run a digit leaves the tile with, but measured slower: see radix_sort.)
// LOW32: last pass of the packed record sort -- only the low half of every key (the arrival index)
// is stored, to pout32, which makes the sorted keys themselves unnecessary.
#ifndef MGX_SORT_PREFETCH_P32
#define MGX_SORT_PREFETCH_P32 1
#endif
constexpr bool kPrefetchP32 = MGX_SORT_PREFETCH_P32 != 0;     // the 32-bit payload is loaded with the keys, not between two barriers
// ONESWEEP: no precomputed offsets.  `goff` then holds the 256 digit totals of this pass, `sweep` points at
// [8 tickets | pad to 64 words | n_tiles x 256 status words], zeroed before the launch:
//   * tile assignment: workgroup b belongs to class x = b & 7 (the XCD it lands on under round-robin placement -- a matter
//     of speed only) and takes ticket k from counter x.  Tiles go in chunks of kSweepChunk consecutive tiles, chunk c to class
//     c & 7, so the two halves of a line that two neighbouring tiles complete still meet in one L2 (15 times out of 16), and
//     the classes run a DIAGONAL schedule: class x starts x chunks late (its first x * kSweepChunk tickets map to no tile and
//     exit), so the chunk before a class's current chunk -- another class's -- was started a whole chunk earlier and has
//     published its prefixes when it is asked, and the tile before a tile inside a chunk holds the ticket just before its
//     own.  Look-back walks stay short (a first version with all classes in step walked up to 8 chunks of counts per tile
//     and lost 0.5 ms per 200 M-key pass);
//   * status: two digits per 64-bit word, each half flag << 30 | count; flag 1 = the tile's own count, 2 = inclusive prefix
//     over all tiles up to it.  A tile publishes its counts as soon as it has ranked its keys, walks back over its
//     predecessors adding counts until it meets a prefix, publishes its own prefix and scatters.  One self-contained word per
//     hand-off: relaxed agent-scope atomics (sc1 accesses), no fences;
//   * every wait is bounded: a predecessor that does not show up within kSweepSpinLimit polls sets the error word, the tile
//     carries on with clamped offsets (no out-of-range store) and the host re-runs the sort with the histogram passes.
constexpr int kSweepChunk = 16;
constexpr u32 kSweepHeaderWords = 64;
constexpr u32 kSweepSpinLimit = 1u << 22;
inline u32 sweep_grid_size(u32 n_tiles) { return (((n_tiles + kSweepChunk - 1) / kSweepChunk + 7) / 8 + 1) * 8 * kSweepChunk; }
template <bool HAS_P64, bool HAS_P32, int WAVES, bool LOW32 = false, bool ONESWEEP = false>
__global__ __launch_bounds__(WAVES * 64) void k_radix_scatter(const u64* __restrict__ kin, u64* __restrict__ kout,
                                                              const u64* __restrict__ pin64, u64* __restrict__ pout64,
                                                              const u32* __restrict__ pin32, u32* __restrict__ pout32,
                                                              u32 n, int shift, const u32* __restrict__ goff, u32 n_tiles, int xcd_order,
                                                              u32* __restrict__ sweep = nullptr, u32* __restrict__ sweep_err = nullptr) {
    constexpr int T = WAVES * 64;           // threads
    constexpr int ITEMS = kItems;           // keys per thread
    constexpr int kTile = T * ITEMS;        // keys per workgroup: 4096 with 4 wavefronts, 8192 with 8
    constexpr int SLICE = kTile / WAVES;    // keys per wavefront
    __shared__ u64 sbuf[kTile];             // exchange buffer (keys, then payloads): 32 KB / 64 KB
    __shared__ u32 wcnt[WAVES][256];        // per wavefront and digit: running count, later the slot base
    __shared__ u32 gdelta[256];             // global offset of the digit's run minus its tile-local start
    __shared__ u32 sm[WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroups are dealt to the 8 XCDs round-robin and every XCD has its own L2.  A digit's run of
    // ~16 keys starts at an arbitrary 8-byte offset, so most 128-byte lines are completed by the
    // NEXT tile: giving each XCD a contiguous range of tiles (walked in order) lets both halves of
    // such a line meet in one L2 instead of reaching HBM as two partial writes from two L2s.
    u32 tile = blockIdx.x;
    if constexpr (ONESWEEP) {
        if (tid == 0) sm[0] = atomicAdd(&sweep[blockIdx.x & 7u], 1u);
        __syncthreads();
        const u32 x = blockIdx.x & 7u, lag = xcd_order > 1 ? x * (kSweepChunk / 8) : 0u;      // (xcd_order 2: staggered classes)
        if (sm[0] < lag) return;                    // (uniform) class x starts x eighths of a chunk late: tickets then follow tile order
        const u32 k = sm[0] - lag;
        tile = ((k / kSweepChunk) * 8u + x) * kSweepChunk + k % kSweepChunk;
        if (tile >= n_tiles) return;                // the grid is rounded up to whole rounds of chunks
    } else if (xcd_order) {
        const u32 q = n_tiles >> 3, r = n_tiles & 7u, x = blockIdx.x & 7u;
        tile = x * q + min(x, r) + (blockIdx.x >> 3);
    }
    const u32 tile0 = tile * kTile;
    const u32 tile_n = min((u32)kTile, n - tile0);
    for (int i = tid; i < WAVES * 256; i += T) (&wcnt[0][0])[i] = 0;
    u32 my_goff = 0;
    if constexpr (!ONESWEEP) my_goff = tid < 256 ? goff[(size_t)tile * 256 + tid] : 0u;
    __syncthreads();

    // phase 1: stable rank of every key inside its wavefront's slice
    u64 key[ITEMS];
    u64 pay[HAS_P64 ? ITEMS : 1];
    u32 lrank[ITEMS];
    constexpr bool KEEP_DIG = !(HAS_P64 && ITEMS > 8);   // register budget: recompute the digit in the widest form
    u32 dig[KEEP_DIG ? ITEMS : 1];
    const u64 lt = lanemask_lt();
    const bool hi_word = shift >= 32;       // digits are byte aligned (shift is a multiple of 8): never straddle
    const int sh = shift & 31;
    // all global loads of the tile are issued up front: the payload's HBM latency is then hidden
    // behind the ranking arithmetic instead of being exposed between two barriers later on.
    // Slots past the end of the input hold ~0: they are the last of the tile in input order and
    // carry the last digit, so ranking them like keys changes no valid key's slot.
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 li = wave * SLICE + k * 64 + lane;            // local index: slices are contiguous per wave
        key[k] = li < tile_n ? kin[tile0 + li] : ~0ull;
    }
    if constexpr (HAS_P64) {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const u32 li = wave * SLICE + k * 64 + lane;
            pay[k] = li < tile_n ? pin64[tile0 + li] : 0ull;
        }
    }
    u32 pay32[(HAS_P32 && kPrefetchP32) ? ITEMS : 1];
    if constexpr (HAS_P32 && kPrefetchP32) {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const u32 li = wave * SLICE + k * 64 + lane;
            pay32[k] = li < tile_n ? pin32[tile0 + li] : 0u;
        }
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 d = ((hi_word ? (u32)(key[k] >> 32) : (u32)key[k]) >> sh) & 255u;
        if constexpr (KEEP_DIG) dig[k] = d;
        u64 peers = ~0ull;                                       // lanes holding the same digit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const u64 m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const u32 r = __popcll(peers & lt);
        u32 old = 0;
        if (r == 0) old = atomicAdd(&wcnt[wave][d], (u32)__popcll(peers));
        old = __shfl(old, __ffsll((long long)peers) - 1, 64);
        lrank[k] = old + r;
    }
    __syncthreads();
    // phase 2: per digit, exclusive over the wavefronts, then exclusive over digits
    {
        u32 c[WAVES], tot = 0;
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) { c[w] = wcnt[w][tid]; tot += c[w]; }
        }
        if constexpr (ONESWEEP) {
            // two digits per 64-bit status word: the even thread of a pair speaks for both
            u64* const status = reinterpret_cast<u64*>(sweep + kSweepHeaderWords);
            const u32 tot_hi = __shfl_down(tot, 1, 64);
            u32 before = 0, before_hi = 0;                       // keys with this digit in the tiles before this one
            if (tid < 256 && !(tid & 1)) {
                u64* const mine = status + (size_t)tile * 128 + (tid >> 1);
                const u64 counts = (u64)tot | (u64)tot_hi << 32;
                constexpr u64 kAgg = 0x4000000040000000ull, kPre = 0x8000000080000000ull, kVal = 0x3FFFFFFF3FFFFFFFull;
                if (tile == 0) __hip_atomic_store(mine, kPre | counts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else {
                    __hip_atomic_store(mine, kAgg | counts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const u64* p = mine - 128;
                    u32 spins = 0;
                    u64 acc = 0;                                 // both halves at once: the counts stay below 2^30, no carry crosses
                    // four predecessors per round trip (the walk is a chain of dependent loads otherwise); tiles below 0 do not
                    // exist: tile 0 always carries a prefix, so the walk ends there at the latest
                    u32 back = tile;                             // predecessors left
                    bool done = false;
                    while (!done) {
                        u64 w[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) w[q] = (u32)q < back ? __hip_atomic_load(p - 128 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (done) break;
                            u64 v = w[q];
                            while ((v & (kAgg | kPre)) == 0) {   // not published yet
                                if (++spins > kSweepSpinLimit) { atomicOr(sweep_err, 1u); v = kPre; break; }
                                __builtin_amdgcn_s_sleep(1);
                                v = __hip_atomic_load(p - 128 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            acc += v & kVal;
                            if (v & kPre) done = true;
                        }
                        p -= 512; back = back > 4 ? back - 4 : 0;
                    }
                    before = (u32)acc & 0x3FFFFFFFu; before_hi = (u32)(acc >> 32) & 0x3FFFFFFFu;
                    __hip_atomic_store(mine, kPre | ((acc + counts) & kVal), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            const u32 from_even = __shfl_up(before_hi, 1, 64);
            if (tid & 1) before = from_even;
            const u32 dbase = block_excl_scan<WAVES>(tid < 256 ? goff[tid] : 0u, sm);     // where the digit starts in the output
            my_goff = dbase + before;
            if (my_goff > n) my_goff = n;                        // only after a timed-out wait: keep every store in range
        }
        const u32 tb = block_excl_scan<WAVES>(tot, sm);          // threads >= 256 contribute 0 after all digits
        if (tid < 256) {
            u32 run = tb;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) { wcnt[w][tid] = run; run += c[w]; }
            gdelta[tid] = my_goff - tb;
        }
    }
    __syncthreads();
    // phase 3: keys to their tile-local sorted slot, then out in runs of consecutive addresses
    u32 lpos[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 li = wave * SLICE + k * 64 + lane;
        const u32 d = KEEP_DIG ? dig[KEEP_DIG ? k : 0] : ((hi_word ? (u32)(key[k] >> 32) : (u32)key[k]) >> sh) & 255u;
        lpos[k] = wcnt[wave][d] + lrank[k];
        if (li < tile_n) sbuf[lpos[k]] = key[k];
    }
    __syncthreads();
    u32 gpos[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const u32 i = k * T + tid;
        gpos[k] = 0;
        if (i < tile_n) {
            const u64 kk = sbuf[i];
            const u32 d = ((hi_word ? (u32)(kk >> 32) : (u32)kk) >> sh) & 255u;
            gpos[k] = gdelta[d] + i;
            if constexpr (ONESWEEP) { if (gpos[k] >= n) gpos[k] = n - 1; }      // unreachable unless a look-back wait timed out
            if constexpr (LOW32) pout32[gpos[k]] = (u32)kk;
            else kout[gpos[k]] = kk;
        }
    }
    if constexpr (HAS_P64) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const u32 li = wave * SLICE + k * 64 + lane;
            if (li < tile_n) sbuf[lpos[k]] = pay[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const u32 i = k * T + tid;
            if (i < tile_n) pout64[gpos[k]] = sbuf[i];
        }
    }
    if constexpr (HAS_P32) {
        __syncthreads();
        u32* sbuf32 = reinterpret_cast<u32*>(sbuf);
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const u32 li = wave * SLICE + k * 64 + lane;
            if (li < tile_n) sbuf32[lpos[k]] = kPrefetchP32 ? pay32[kPrefetchP32 ? k : 0] : pin32[tile0 + li];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const u32 i = k * T + tid;
            if (i < tile_n) pout32[gpos[k]] = sbuf32[i];
        }
    }
}

// packed coordinate keys (coord << 32 | arrival index): the order is the low half of the sorted keys
__global__ __launch_bounds__(256) void k_unpack_order(const u64* __restrict__ keys, u32 n, u32* __restrict__ order) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) order[i] = (u32)keys[i];
}

// double_pair_indicator (main.cpp:181-192), set from SORTED pair entries so that consecutive lanes
// touch consecutive words: END = 2 after the sort by mate 5' end (sets the record-2 bits), END = 1
// after the sort by sort_key (sets the record-1 bits).  Random-order atomics into the 4L-bit map
// cost a 64-byte read-modify-write each; in sorted order they stay in L2.
template <int END, bool PK>
__global__ __launch_bounds__(256) void k_set_indicator(const u64* __restrict__ k1, const u64* __restrict__ k2, u32 n,
                                                       u32* __restrict__ indicator, u64 indicator_bits, u64 L) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u64 a1 = k1[i];
    const u32 orient = (u32)(a1 & 3);
    u64 b;
    if (END == 2) b = (PK ? (k2[i] >> 32) : k2[i]) + ((orient == 0u || orient == 2u) ? 0ull : L);   // record 2 forward: FF, RF
    else          b = (a1 >> 2) + ((orient == 0u || orient == 1u) ? 0ull : L);   // record 1 forward: FF, FR
    if (b < indicator_bits) atomicOr(&indicator[b >> 5], 1u << (b & 31));
}

// Tiled form of the same bitmap, used whenever every 5' end lies below L - 64 (always, except for
// positions wrapped below zero or clipped past the last contig): the reverse-strand half then
// starts at Lp = L rounded up to a whole tile instead of L -- no forward bit can reach it, so the
// answers are unchanged -- and each workgroup OWNS kIndTile positions of both halves: it finds its
// slice of the sorted entries by binary search, sets bits in LDS and writes whole words with plain
// coalesced stores.  No global atomics and no memset of the 4L-bit map.
constexpr u32 kIndTile = 65536;              // positions per workgroup (2 x 8 KB of LDS)
template <int END, bool PK, bool DEFINE>
__global__ __launch_bounds__(256) void k_indicator_tiles(const u64* __restrict__ k1, const u64* __restrict__ k2, u32 n,
                                                         u32* __restrict__ indicator, u64 Lp) {
    __shared__ u32 fw[kIndTile / 32], rv[kIndTile / 32];
    __shared__ u32 s_lo, s_hi;
    auto POS = [&](u32 t) -> u64 { return END == 2 ? (PK ? (k2[t] >> 32) : k2[t]) : (k1[t] >> 2); };
    for (u32 w = threadIdx.x; w < kIndTile / 32; w += 256) { fw[w] = 0; rv[w] = 0; }
    const u64 pos_lo = (u64)blockIdx.x * kIndTile, pos_hi = pos_lo + kIndTile;
    if (threadIdx.x < 2) {
        const u64 target = threadIdx.x == 0 ? pos_lo : pos_hi;
        u32 a = 0, b = n;                                   // first entry with POS >= target
        while (a < b) { const u32 m = a + (b - a) / 2; if (POS(m) < target) a = m + 1; else b = m; }
        if (threadIdx.x == 0) s_lo = a; else s_hi = a;
    }
    __syncthreads();
    const u32 lo = s_lo, hi = s_hi;
    for (u32 i = lo + threadIdx.x; i < hi; i += 256) {
        const u32 orient = (u32)(k1[i] & 3);
        const u32 p = (u32)(POS(i) - pos_lo);
        const bool fwd = END == 2 ? (orient == 0u || orient == 2u) : (orient == 0u || orient == 1u);
        atomicOr(fwd ? &fw[p >> 5] : &rv[p >> 5], 1u << (p & 31));
    }
    __syncthreads();
    u32* gf = indicator + (size_t)blockIdx.x * (kIndTile / 32);
    u32* gr = indicator + (size_t)(Lp >> 5) + (size_t)blockIdx.x * (kIndTile / 32);
    for (u32 w = threadIdx.x; w < kIndTile / 32; w += 256) {
        if (DEFINE) { gf[w] = fw[w]; gr[w] = rv[w]; }                         // first pass: defines every word
        else { if (fw[w]) gf[w] |= fw[w]; if (rv[w]) gr[w] |= rv[w]; }        // later passes: this tile owns them
    }
}

// Near pairs (one key word p1 << 32 | orient << 30 | delta << 16 | inverted score, sorted on its upper 48 bits): both ends of every pair in ONE
// pass.  A pair's record-1 end lies in the tile of p1, its record-2 end at p1 + delta < p1 + kNearSpan in
// that tile or the next, so tile t scans the entries with p1 in [t*65536 - (kNearSpan - 1), (t+1)*65536).
// Always the first pass over the bitmap: defines every word.
// sub_start[x] = first sorted entry whose record-1 end is at or beyond x * kNearSpan (written by k_find_runs):
// the tile's slice of the entries without two 27-step binary searches per workgroup.
__global__ __launch_bounds__(256) void k_indicator_tiles_near(const u64* __restrict__ nk, u32 n, u32* __restrict__ indicator, u64 Lp,
                                                              const u32* __restrict__ sub_start, u32 n_sub) {
    __shared__ u32 fw[kIndTile / 32], rv[kIndTile / 32];
    __shared__ u32 s_lo, s_hi;
    for (u32 w = threadIdx.x; w < kIndTile / 32; w += 256) { fw[w] = 0; rv[w] = 0; }
    const u64 pos_lo = (u64)blockIdx.x * kIndTile, pos_hi = pos_lo + kIndTile;
    constexpr u32 kSubPerTile = kIndTile / (u32)kNearSpan;
    if (threadIdx.x == 0) {
        // entries with p1 >= pos_lo - kNearSpan (a superset of the pairs whose second end can reach this tile) ...
        s_lo = blockIdx.x == 0 ? 0u : sub_start[blockIdx.x * kSubPerTile - 1];
        // ... up to the first entry at or beyond pos_hi
        s_hi = sub_start[min((blockIdx.x + 1) * kSubPerTile, n_sub)];
    }
    __syncthreads();
    const u32 lo = s_lo, hi = min(s_hi, n);
    for (u32 i = lo + threadIdx.x; i < hi; i += 256) {
        const u64 key = nk[i];
        const u64 p1 = key >> kNearShift, p2 = p1 + ((key >> kNearDeltaShift) & (kNearSpan - 1));
        const u32 orient = (u32)(key >> kNearOrientShift) & 3u;
        if (p1 >= pos_lo) {                                  // record 1 forward: FF, FR
            const u32 p = (u32)(p1 - pos_lo);
            atomicOr((orient == 0u || orient == 1u) ? &fw[p >> 5] : &rv[p >> 5], 1u << (p & 31));
        }
        if (p2 >= pos_lo && p2 < pos_hi) {                   // record 2 forward: FF, RF
            const u32 p = (u32)(p2 - pos_lo);
            atomicOr((orient == 0u || orient == 2u) ? &fw[p >> 5] : &rv[p >> 5], 1u << (p & 31));
        }
    }
    __syncthreads();
    u32* gf = indicator + (size_t)blockIdx.x * (kIndTile / 32);
    u32* gr = indicator + (size_t)(Lp >> 5) + (size_t)blockIdx.x * (kIndTile / 32);
    for (u32 w = threadIdx.x; w < kIndTile / 32; w += 256) { gf[w] = fw[w]; gr[w] = rv[w]; }
}

// atomic fallback for near pairs (reference bitmap layout)
__global__ __launch_bounds__(256) void k_set_indicator_near(const u64* __restrict__ nk, u32 n, u32* __restrict__ indicator,
                                                            u64 indicator_bits, u64 L) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u64 key = nk[i];
    const u64 p1 = key >> kNearShift, p2 = p1 + ((key >> kNearDeltaShift) & (kNearSpan - 1));
    const u32 orient = (u32)(key >> kNearOrientShift) & 3u;
    const u64 b1 = p1 + ((orient == 0u || orient == 1u) ? 0ull : L), b2 = p2 + ((orient == 0u || orient == 2u) ? 0ull : L);
    if (b1 < indicator_bits) atomicOr(&indicator[b1 >> 5], 1u << (b1 & 31));
    if (b2 < indicator_bits) atomicOr(&indicator[b2 >> 5], 1u << (b2 & 31));
}

// ---------------------------------------------------------------------------------------------
// best-of-run and duplicate marking
// ---------------------------------------------------------------------------------------------
// both records of a losing pair: one 2-byte store when the mate is the neighbour (flag in the record word)
__device__ __forceinline__ void mark_pair(uint8_t* dup, const mgx_rec_t* recs, u32 r, bool neighbour) {
    if (neighbour) reinterpret_cast<uint16_t*>(dup)[r >> 1] = 0x0101;
    else { dup[r] = 1; dup[recs[r].mate] = 1; }
}

// quality of a pair entry: smaller is better -- score descending, then tile, x, y ascending
// (main.cpp:253-264 / 303-314); the record index (= arrival order) breaks total ties
__device__ __forceinline__ u64 quality_word(uint16_t score, const mgx_rec_t& a) {
    return ((u64)(0xFFFFu - (u32)score) << 48) | ((u64)a.tile << 32) | ((u64)a.x << 16) | (u64)a.y;
}
__device__ __forceinline__ u64 quality_double(const mgx_rec_t* recs, u32 rec) {
    const mgx_rec_t a = recs[rec];
    const mgx_rec_t b = recs[a.mate];
    const u32 score = (u32)(uint16_t)(a.score + b.score);                // pair.cpp:81
    return ((u64)(0xFFFFu - score) << 48) | ((u64)a.tile << 32) | ((u64)a.x << 16) | (u64)a.y;
}
__device__ __forceinline__ u64 quality_single(const mgx_rec_t* recs, u32 rec) {
    const mgx_rec_t a = recs[rec];
    return ((u64)(0xFFFFu - (u32)a.score) << 48) | ((u64)a.tile << 32) | ((u64)a.x << 16) | (u64)a.y;
}

// Duplicate search over the sorted entries, in two steps so that the random-access part runs dense:
//   k_find_runs : one lane per entry finds the run heads.  A run of ONE entry (nine in ten) needs no
//                 quality comparison at all: nothing to mark for pairs; for singles only the
//                 indicator test of main.cpp:325-331.  Heads of longer runs are appended to a list
//                 (LDS slots + one global atomic per 4096-entry workgroup; list order is irrelevant).
//   k_mark_list : one lane per listed head walks its run (runs are short), gathers the records'
//                 quality words and marks everything but the best.  Runs longer than kWalkCap go to
//                 a second list that k_mark_long handles with a whole workgroup each.
// PK: the second array holds (mate 5' end << 32 | record) packed in one word and `rec` is unused.
constexpr int kFindItems = 16;

template <bool DOUBLE, bool PK, int KS>
__global__ __launch_bounds__(256) void k_find_runs(const u64* __restrict__ k1, const u64* __restrict__ k2,
                                                   const u32* __restrict__ rec, u32 n,
                                                   const u32* __restrict__ indicator, u64 indicator_bits, u64 L,
                                                   uint8_t* __restrict__ dup, u32* __restrict__ multi_list, u32* n_multi,
                                                   u32* __restrict__ sub_start, u32 n_sub) {
    __shared__ u32 s_base;
    __shared__ u32 s_scan[4];
    // near pairs come as DOUBLE with k2 == nullptr: the single key word is the whole identity
    auto K2 = [&](u32 t) -> u64 { return (!DOUBLE || !k2) ? 0ull : (PK ? (k2[t] >> 32) : k2[t]); };
    auto REC = [&](u32 t) -> u32 { return PK ? (u32)k2[t] : rec[t]; };
    const u32 base = blockIdx.x * (256 * kFindItems);
    u32 is_multi = 0;
#pragma unroll
    for (int k = 0; k < kFindItems; ++k) {
        const u32 i = base + k * 256 + threadIdx.x;
        bool multi = false;
        // neighbours come from the adjacent lanes; only the two edge lanes of a wavefront load theirs
        const int lane = threadIdx.x & 63;
        const u64 a1 = i < n ? k1[i] >> KS : 0ull, a2 = i < n ? K2(i) : 0ull;     // KS: key bits below the identity
        u64 p1 = __shfl_up(a1, 1, 64), p2 = DOUBLE ? __shfl_up(a2, 1, 64) : 0ull;
        u64 n1 = __shfl_down(a1, 1, 64), n2 = DOUBLE ? __shfl_down(a2, 1, 64) : 0ull;
        if (lane == 0 && i > 0 && i < n) { p1 = k1[i - 1] >> KS; p2 = K2(i - 1); }
        if (lane == 63 && i + 1 < n) { n1 = k1[i + 1] >> KS; n2 = K2(i + 1); }
        if constexpr (KS >= kNearScoreBits) {
            // near pairs: where the sorted entries cross a kNearSpan boundary of the genome (for the bitmap pass)
            if (sub_start && i < n) {
                const u32 cur = (u32)((a1 >> (kNearShift - KS)) >> kNearDeltaBits);
                const int prev = i == 0 ? -1 : (int)((p1 >> (kNearShift - KS)) >> kNearDeltaBits);
                for (int x = prev + 1; x <= (int)cur && x <= (int)n_sub; ++x) sub_start[x] = i;
                if (i == n - 1) for (u32 x = cur + 1; x <= n_sub; ++x) sub_start[x] = n;
            }
        }
        if (i < n) {
            const bool head = i == 0 || p1 != a1 || (DOUBLE && p2 != a2);
            if (head) {
                multi = i + 1 < n && n1 == a1 && (!DOUBLE || n2 == a2);
                if (!DOUBLE && !multi) {
                    // main.cpp:325-331: the kept single is a duplicate iff a double pair has an end there
                    const u64 target = (a1 >> 2) + (((a1 & 3) == 3) ? L : 0ull);
                    if (target < indicator_bits && ((indicator[target >> 5] >> (target & 31)) & 1u)) dup[REC(i)] = 1;
                }
            }
        }
        is_multi |= (multi ? 1u : 0u) << k;
    }
    // one block scan over the per-thread counts (instead of a ballot + LDS atomic per item), one global
    // atomic per workgroup; the list's order is irrelevant
    u32 total;
    u32 at = block_excl_scan_256((u32)__popc(is_multi), s_scan, &total);
    if (threadIdx.x == 0) s_base = total ? atomicAdd(n_multi, total) : 0u;
    __syncthreads();
    at += s_base;
#pragma unroll
    for (int k = 0; k < kFindItems; ++k)
        if ((is_multi >> k) & 1u) multi_list[at++] = base + k * 256 + threadIdx.x;
}

template <bool DOUBLE, bool PK, int KS>
__global__ __launch_bounds__(256) void k_mark_list(const u64* __restrict__ k1, const u64* __restrict__ k2,
                                                   const u32* __restrict__ rec, u32 n, const mgx_rec_t* __restrict__ recs, u32 n_records,
                                                   const u32* __restrict__ indicator, u64 indicator_bits, u64 L,
                                                   uint8_t* __restrict__ dup, const u32* __restrict__ multi_list, const u32* n_multi,
                                                   u32* __restrict__ long_list, u32* n_long, u32 rec_mask) {
    // rec_mask: near-pair record words carry "the mate is the neighbour" in the bits outside the mask
    auto K2 = [&](u32 t) -> u64 { return (!DOUBLE || !k2) ? 0ull : (PK ? (k2[t] >> 32) : k2[t]); };
    auto REC = [&](u32 t) -> u32 { return PK ? (u32)k2[t] : rec[t] & rec_mask; };
    auto MATE = [&](u32 t, u32 r) -> u32 { return (!PK && (rec[t] & ~rec_mask)) ? (r ^ 1u) : recs[r].mate; };
    const u32 total = *n_multi;
    for (u32 li = blockIdx.x * 256 + threadIdx.x; li < total; li += gridDim.x * 256) {
        const u32 i = multi_list[li];
        if constexpr (KS > kNearScoreBits) {
            // near pairs sorted on record 1's 5' end only: the run holds every pair that starts here; an
            // entry loses to any other entry of the run with the same identity (bits above the score) and
            // a better score -- tile, x, y, then arrival order on a score tie
            const u64 run_id = k1[i] >> KS;
            u32 j = i + 1;
            for (; j < n && j - i < kWalkCap; ++j) if ((k1[j] >> KS) != run_id) break;
            if (j < n && j - i >= kWalkCap && (k1[j] >> KS) == run_id) {
                long_list[atomicAdd(n_long, 1u)] = i;                                 // long run: defer
                continue;
            }
            for (u32 t = i; t < j; ++t) {
                const u64 kt = k1[t];
                const u64 idt = kt >> kNearScoreBits;
                const u32 st = (u32)(kt & 0xFFFFu);
                bool loser = false, have_t = false;
                u32 rt = 0; u64 qt = 0;
                for (u32 u = i; u < j && !loser; ++u) {
                    if (u == t) continue;
                    const u64 ku = k1[u];
                    if ((ku >> kNearScoreBits) != idt) continue;
                    const u32 su = (u32)(ku & 0xFFFFu);
                    if (su < st) loser = true;
                    else if (su == st) {
                        if (!have_t) { rt = REC(t); const mgx_rec_t a = recs[rt]; qt = ((u64)a.tile << 32) | ((u64)a.x << 16) | (u64)a.y; have_t = true; }
                        const u32 ru = REC(u);
                        const mgx_rec_t b = recs[ru];
                        const u64 qu = ((u64)b.tile << 32) | ((u64)b.x << 16) | (u64)b.y;
                        if (qu < qt || (qu == qt && ru < rt)) loser = true;
                    }
                }
                if (loser) {
                    if (!have_t) rt = REC(t);
                    mark_pair(dup, recs, rt, !PK && (rec[t] & ~rec_mask));
                }
            }
            continue;
        } else if constexpr (KS > 0) {
            // near pairs: the low KS bits of the key word are 0xFFFF - pair score, so the best entry of
            // the run is known from the sorted keys; records are gathered only when several entries
            // share the best score (tile, x, y, then arrival order decide) and for the losers' mates
            const u64 id = k1[i] >> KS;
            u32 best = i, bs = (u32)(k1[i] & ((1u << KS) - 1));
            bool tie = false;
            u32 j = i + 1;
            for (; j < n && j - i < kWalkCap; ++j) {
                const u64 kj = k1[j];
                if ((kj >> KS) != id) break;
                const u32 sj = (u32)(kj & ((1u << KS) - 1));
                if (sj < bs) { bs = sj; best = j; tie = false; }
                else if (sj == bs) tie = true;
            }
            if (j < n && j - i >= kWalkCap && (k1[j] >> KS) == id) {
                long_list[atomicAdd(n_long, 1u)] = i;                                 // long run: defer
                continue;
            }
            if (tie) {
                u64 bq = ~0ull; u32 br = 0xFFFFFFFFu;
                for (u32 t = i; t < j; ++t) {
                    if ((u32)(k1[t] & ((1u << KS) - 1)) != bs) continue;
                    const u32 rt = REC(t);
                    const mgx_rec_t a = recs[rt];
                    const u64 q = ((u64)a.tile << 32) | ((u64)a.x << 16) | (u64)a.y;
                    if (q < bq || (q == bq && rt < br)) { bq = q; br = rt; best = t; }
                }
            }
            for (u32 t = i; t < j; ++t) {
                if (t == best) continue;
                const u32 r = REC(t);
                mark_pair(dup, recs, r, !PK && (rec[t] & ~rec_mask));
            }
            continue;
        }
        const u64 a1 = k1[i], a2 = K2(i);
        // the first two entries belong to the run by construction: their record gathers (the slow,
        // random part) are issued together instead of one after the other
        const u32 r0 = REC(i), r1 = REC(i + 1);
        const mgx_rec_t ra = recs[r0], rb = recs[r1];
        u64 q0, q1;
        if (DOUBLE) {
            // mates sit next to each other in arrival order (mgx_sortdedup_pack emits record 1, record 2):
            // the neighbour in the same 64-byte pair is fetched together with the record instead of
            // after it; only a mate that lives elsewhere costs a second, dependent gather
            const u32 na = r0 ^ 1u, nb = r1 ^ 1u;
            uint16_t ma = na < n_records ? recs[na].score : (uint16_t)0, mb = nb < n_records ? recs[nb].score : (uint16_t)0;
            if (ra.mate != na) ma = recs[ra.mate].score;
            if (rb.mate != nb) mb = recs[rb.mate].score;
            q0 = quality_word((uint16_t)(ra.score + ma), ra);                 // pair.cpp:81: uint16 sum
            q1 = quality_word((uint16_t)(rb.score + mb), rb);
        } else {
            q0 = quality_word(ra.score, ra); q1 = quality_word(rb.score, rb);
        }
        u32 best = i, best_rec = r0;
        u64 bq = q0;
        if (q1 < bq || (q1 == bq && r1 < best_rec)) { bq = q1; best = i + 1; best_rec = r1; }
        u32 j = i + 2;
        for (; j < n && j - i < kWalkCap; ++j) {
            if (k1[j] != a1 || (DOUBLE && K2(j) != a2)) break;
            const u32 rj = REC(j);
            const u64 q = DOUBLE ? quality_double(recs, rj) : quality_single(recs, rj);
            if (q < bq || (q == bq && rj < best_rec)) { bq = q; best = j; best_rec = rj; }
        }
        if (j < n && j - i >= kWalkCap && k1[j] == a1 && (!DOUBLE || K2(j) == a2)) {
            long_list[atomicAdd(n_long, 1u)] = i;                                     // long run: defer
            continue;
        }
        if (!DOUBLE) {
            const u64 target = (a1 >> 2) + (((a1 & 3) == 3) ? L : 0ull);
            if (target < indicator_bits && ((indicator[target >> 5] >> (target & 31)) & 1u)) dup[REC(best)] = 1;
        }
        for (u32 t = i; t < j; ++t) {
            if (t == best) continue;
            const u32 r = REC(t);
            dup[r] = 1;
            if (DOUBLE) dup[MATE(t, r)] = 1;
        }
    }
}

template <bool DOUBLE, bool PK, int KS>
__global__ __launch_bounds__(256) void k_mark_long(const u64* __restrict__ k1, const u64* __restrict__ k2,
                                                   const u32* __restrict__ rec, u32 n, const mgx_rec_t* __restrict__ recs,
                                                   const u32* __restrict__ indicator, u64 indicator_bits, u64 L,
                                                   uint8_t* __restrict__ dup, const u32* __restrict__ long_list, const u32* n_long, u32 rec_mask) {
    __shared__ u64 sq[256];
    __shared__ u32 sp[256];
    __shared__ u32 sr[256];
    __shared__ u32 s_end;
    auto K2 = [&](u32 t) -> u64 { return (!DOUBLE || !k2) ? 0ull : (PK ? (k2[t] >> 32) : k2[t]); };
    auto REC = [&](u32 t) -> u32 { return PK ? (u32)k2[t] : rec[t] & rec_mask; };
    for (u32 li = blockIdx.x; li < *n_long; li += gridDim.x) {
        const u32 i = long_list[li];
        const u64 a1 = k1[i] >> KS, a2 = K2(i);          // KS: key bits below the identity
        // pass 1: extent of the run and its best entry
        u64 bq = ~0ull; u32 bp = 0xFFFFFFFFu, bpr = 0xFFFFFFFFu;
        u32 end = n;
        for (u32 c = i; c < n; c += 256) {
            const u32 t = c + threadIdx.x;
            const bool in = t < n && (k1[t] >> KS) == a1 && (!DOUBLE || K2(t) == a2);
            if (threadIdx.x == 0) s_end = n;
            __syncthreads();
            if (t < n && !in) atomicMin(&s_end, t);
            __syncthreads();
            const u32 e = s_end;
            if (t < e) {
                const u64 q = DOUBLE ? quality_double(recs, REC(t)) : quality_single(recs, REC(t));
                if (q < bq || (q == bq && REC(t) < bpr)) { bq = q; bp = t; bpr = REC(t); }
            }
            __syncthreads();
            if (e < n) { end = e; break; }
        }
        sq[threadIdx.x] = bq; sp[threadIdx.x] = bp; sr[threadIdx.x] = bpr;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                const u64 q = sq[threadIdx.x + s]; const u32 p = sp[threadIdx.x + s], r = sr[threadIdx.x + s];
                if (q < sq[threadIdx.x] || (q == sq[threadIdx.x] && r < sr[threadIdx.x])) { sq[threadIdx.x] = q; sp[threadIdx.x] = p; sr[threadIdx.x] = r; }
            }
            __syncthreads();
        }
        const u32 best = sp[0];
        __syncthreads();
        if (!DOUBLE && threadIdx.x == 0) {
            u64 target = (a1 >> 2) + (((a1 & 3) == 3) ? L : 0ull);
            if (target < indicator_bits && ((indicator[target >> 5] >> (target & 31)) & 1u)) dup[REC(best)] = 1;
        }
        for (u32 t = i + threadIdx.x; t < end; t += 256) {
            if (t == best) continue;
            const u32 r = REC(t);
            dup[r] = 1;
            if (DOUBLE) dup[(!PK && (rec[t] & ~rec_mask)) ? (r ^ 1u) : recs[r].mate] = 1;
        }
        __syncthreads();
    }
}

// long runs of the position-sorted near pairs: one workgroup per run, every entry against every other
// (bounded by kHugeRun; beyond that the flag sends the whole pipeline to the six-pass sort)
template <int RS>
__global__ __launch_bounds__(256) void k_mark_long_sub(const u64* __restrict__ k1, const u32* __restrict__ rec, u32 n,
                                                       const mgx_rec_t* __restrict__ recs, uint8_t* __restrict__ dup,
                                                       const u32* __restrict__ long_list, const u32* n_long, u32* huge_runs, u32 rec_mask) {
    __shared__ u32 s_end;
    for (u32 li = blockIdx.x; li < *n_long; li += gridDim.x) {
        const u32 i = long_list[li];
        const u64 run_id = k1[i] >> RS;
        u32 end = n;
        for (u32 c = i; c < n; c += 256) {                     // extent of the run
            const u32 t = c + threadIdx.x;
            if (threadIdx.x == 0) s_end = n;
            __syncthreads();
            if (t < n && (k1[t] >> RS) != run_id) atomicMin(&s_end, t);
            __syncthreads();
            const u32 e = s_end;
            __syncthreads();
            if (e < n) { end = e; break; }
        }
        if (end - i > kHugeRun) {
            if (threadIdx.x == 0) atomicOr(huge_runs, 1u);
            continue;
        }
        for (u32 t = i + threadIdx.x; t < end; t += 256) {
            const u64 kt = k1[t];
            const u64 idt = kt >> kNearScoreBits;
            const u32 st = (u32)(kt & 0xFFFFu);
            bool loser = false, have_t = false;
            u32 rt = 0; u64 qt = 0;
            for (u32 u = i; u < end && !loser; ++u) {
                if (u == t) continue;
                const u64 ku = k1[u];
                if ((ku >> kNearScoreBits) != idt) continue;
                const u32 su = (u32)(ku & 0xFFFFu);
                if (su < st) loser = true;
                else if (su == st) {
                    if (!have_t) { rt = rec[t] & rec_mask; const mgx_rec_t a = recs[rt]; qt = ((u64)a.tile << 32) | ((u64)a.x << 16) | (u64)a.y; have_t = true; }
                    const u32 ru = rec[u] & rec_mask;
                    const mgx_rec_t b = recs[ru];
                    const u64 qu = ((u64)b.tile << 32) | ((u64)b.x << 16) | (u64)b.y;
                    if (qu < qt || (qu == qt && ru < rt)) loser = true;
                }
            }
            if (loser) {
                if (!have_t) rt = rec[t] & rec_mask;
                mark_pair(dup, recs, rt, (rec[t] & ~rec_mask) != 0);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_count_dup(const uint8_t* dup, u32 n, Scalars* sc) {
    // 16 flag bytes per load (the array comes from hipMalloc: 256-byte aligned); bytes are 0 or 1
    u32 c = 0;
    const u32 n16 = n / 16;
    const uint4* d4 = reinterpret_cast<const uint4*>(dup);
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) {
        const uint4 v = d4[i];
        c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    for (u32 i = n16 * 16 + blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) c += dup[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&sc->n_dup, c);
}

inline int bits_of(uint64_t v) { int b = 0; while (v) { ++b; v >>= 1; } return b ? b : 1; }

}  // namespace

struct mgx_sortdedup {
    int device = 0;
    unsigned flags = 0;
    hipStream_t compute = nullptr, copy = nullptr;
    int n_cu = 256;
    uint64_t L = 0;
    u32 n = 0;                             // records the duplicate search works on (the marking half of a shard)
    u32 n_order = 0;                       // records the coordinate sort orders (== n unless the input is a shard)
    bool sharded = false;
    u64* d_ocoord = nullptr; u32* d_oarr = nullptr; size_t order_cap = 0;   // a shard's ordering half as uploaded
    u64* d_marks = nullptr; size_t marks_cap = 0; u32 n_marks = 0;          // indicator marks routed from other shards
    size_t cap = 0;                        // record capacity of the work buffers below
    mgx_rec_t* d_recs = nullptr; size_t recs_cap = 0;      // the uploaded records (grown on demand by a streamed upload)
    void* d_wire[2] = {nullptr, nullptr};  // device side of the staging buffers: compact 24-byte records before expansion
    uint64_t up_L = 0, up_high = 0;        // streamed upload in progress: reference length, highest record index seen + 1
    bool uploading = false;
    u64 *d_ckey[2] = {nullptr, nullptr}; u32* d_cval[2] = {nullptr, nullptr};
    u64 *d_k1[2] = {nullptr, nullptr}, *d_k2[2] = {nullptr, nullptr}; u32* d_prec[2] = {nullptr, nullptr};
    u64* d_sk1[2] = {nullptr, nullptr}; u32* d_srec[2] = {nullptr, nullptr};
    u64* d_nk[2] = {nullptr, nullptr}; u32* d_nrec[2] = {nullptr, nullptr};     // near double pairs
    // three independent sorts run concurrently (main: far pairs + singles, side[0]: near pairs,
    // side[1]: records), each with its own histogram / scan / long-run scratch
    struct Scratch { u32 *hist = nullptr, *chunk = nullptr, *longl = nullptr, *multi = nullptr, *totals = nullptr; } scr[3];
    bool onesweep_failed = false;          // a look-back wait timed out once: this context keeps to the histogram passes
    bool onesweep = false;                 // decoupled look-back scatters (k_radix_scatter<..., ONESWEEP>); off after a timed-out wait or MGX_SORTDEDUP_ONESWEEP=0
    hipStream_t side[2] = {nullptr, nullptr};
    hipEvent_t ev_ind = nullptr, ev_side[2] = {nullptr, nullptr};
    u32* d_indicator = nullptr; uint64_t indicator_bits = 0; size_t indicator_cap_words = 0;
    u32* d_sub_start = nullptr; size_t sub_cap = 0;      // first near entry per kNearSpan positions (bitmap pass)
    uint8_t* d_dup = nullptr;
    u32* d_half_hist = nullptr;            // [ceil(n / kBuildBlock)][256], written by the build kernel
    Scalars* d_sc = nullptr;
    void* pinned[2] = {nullptr, nullptr};
    size_t pinned_cap = 0;
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    std::vector<hipEvent_t> ev_scatter;    // pairs (start, stop) per scatter launch
    size_t ev_used = 0;
    size_t ev_rec_begin = 0, ev_rec_end = 0;   // event range of the record sort's scatter launches
    uint64_t rec_scatter_bytes = 0;
    uint64_t scatter_bytes = 0;
    int order_buf = 0;                     // which d_cval holds the final order
    bool packed_coord = false;             // L < 2^32: coordinate sort on packed (coord, index) words
    bool packed_pair = false;              // every 5' end < 2^32: (mate end, record) ride in one word
    bool ran = false;
    bool near_by_position = true;          // near pairs sorted on record 1's 5' end only (MGX_SORTDEDUP_NEAR_EXACT=1: six-pass sort)
    bool finished = false;                 // the results of the last run have been checked for the fallback
    int wide_tiles = -1;                   // MGX_SORTDEDUP_WIDE_TILES=1: 8192-key scatter tiles (measured slower; default 4096)
    int xcd_order = 1;                     // scatter tiles walk each XCD's contiguous range (MGX_SORTDEDUP_XCD_ORDER=0: plain)
    Scalars sc{};
    mgx_sortdedup_stats_t stats{};
};

namespace {

constexpr size_t kPinnedChunk = 64u << 20;

void free_buffers(mgx_sortdedup* c) {
    for (int i = 0; i < 2; ++i) {
        (void)hipFree(c->d_ckey[i]); (void)hipFree(c->d_cval[i]); (void)hipFree(c->d_k1[i]); (void)hipFree(c->d_k2[i]);
        (void)hipFree(c->d_prec[i]); (void)hipFree(c->d_sk1[i]); (void)hipFree(c->d_srec[i]);
        (void)hipFree(c->d_nk[i]); (void)hipFree(c->d_nrec[i]); c->d_nk[i] = nullptr; c->d_nrec[i] = nullptr;
        c->d_ckey[i] = c->d_k1[i] = c->d_k2[i] = c->d_sk1[i] = nullptr;
        c->d_cval[i] = c->d_prec[i] = c->d_srec[i] = nullptr;
    }
    for (auto& q : c->scr) { (void)hipFree(q.hist); (void)hipFree(q.chunk); (void)hipFree(q.longl); (void)hipFree(q.multi); (void)hipFree(q.totals); q.hist = q.chunk = q.longl = q.multi = q.totals = nullptr; }
    (void)hipFree(c->d_dup); c->d_dup = nullptr;
    (void)hipFree(c->d_half_hist); c->d_half_hist = nullptr;
    c->cap = 0;
}

template <typename T>
int dalloc(T** p, size_t count) {
    HIP_TRY(hipMalloc((void**)p, std::max<size_t>(count, 1) * sizeof(T)));
    return 0;
}

int ensure_capacity(mgx_sortdedup* c, size_t n) {
    if (n <= c->cap) return 0;
    free_buffers(c);
    int rc = 0;
    const size_t half = n / 2 + 1;
    const size_t n_tiles = (n + kTile - 1) / kTile + 1;
    for (int i = 0; i < 2 && !rc; ++i) {
        rc |= dalloc(&c->d_ckey[i], n); rc |= dalloc(&c->d_cval[i], n);
        rc |= dalloc(&c->d_k1[i], half); rc |= dalloc(&c->d_k2[i], half); rc |= dalloc(&c->d_prec[i], half);
        rc |= dalloc(&c->d_sk1[i], n); rc |= dalloc(&c->d_srec[i], n);
        rc |= dalloc(&c->d_nk[i], half); rc |= dalloc(&c->d_nrec[i], half);
    }
    for (auto& q : c->scr) {
        rc |= dalloc(&q.hist, n_tiles * 256 + kSweepHeaderWords);      // tile histograms / offsets, or the look-back status words
        rc |= dalloc(&q.totals, kMaxPasses * 256 + 64);                // digit totals of every pass of a sort + the error word
        rc |= dalloc(&q.chunk, ((n_tiles + kChunkTiles - 1) / kChunkTiles + 1) * 256);
        rc |= dalloc(&q.longl, n / kWalkCap + 16);
        rc |= dalloc(&q.multi, n / 2 + 16);          // heads of runs with >= 2 entries
    }
    rc |= dalloc(&c->d_dup, n);
    rc |= dalloc(&c->d_half_hist, (n / kBuildBlock + 2) * 256);
    if (rc) { free_buffers(c); return -ENOMEM; }
    c->cap = n;
    return 0;
}

// one stable LSD radix sort of (key, [p64], p32) over `bits` low bits of the key
// buffers are ping-pong pairs; *cur is the index of the input buffer and is updated
template <int WAVES>
int radix_sort_w(mgx_sortdedup* c, hipStream_t s, const mgx_sortdedup::Scratch& q, u64* key[2], u64* p64[2], u32* p32[2], u32 n,
                 int first_shift, int bits, int* cur, u32* low32_out, bool* low32_done, const u32* first_half_hist) {
    constexpr int TILE = WAVES * 64 * kItems;
    if (low32_done) *low32_done = false;
    if (n == 0) return 0;
    const u32 n_tiles = (n + TILE - 1) / TILE;
    const u32 n_chunks = (n_tiles + kChunkTiles - 1) / kChunkTiles;
    const int n_passes = (bits + 7) / 8;
    const bool sweep = c->onesweep && n < (1u << 30) && n_passes <= kMaxPasses;
    u32* const sweep_err = q.totals + kMaxPasses * 256;
    const u32 sweep_grid = sweep_grid_size(n_tiles);
    const char* env_lag = getenv("MGX_SORTDEDUP_SWEEP_LAG");
    const bool sweep_lag = env_lag ? atoi(env_lag) != 0 : false;
    if (sweep) {
        // every digit of every pass counted in ONE read of the keys
        HIP_TRY(hipMemsetAsync(q.totals, 0, kMaxPasses * 256 * sizeof(u32), s));
        hipLaunchKernelGGL(k_radix_totals, dim3(std::min<u32>(kTotalsBlocks, n_tiles)), dim3(256), 0, s, key[*cur], n, first_shift, n_passes, q.totals);
        c->stats.n_key_hist_launches++;
    }
    for (int shift = first_shift; shift < first_shift + bits; shift += 8) {
        const int in = *cur, out = in ^ 1;
        const u32* goff = q.hist;
        u32 grid = n_tiles;
        if (sweep) {
            HIP_TRY(hipMemsetAsync(q.hist, 0, ((size_t)n_tiles * 256 + kSweepHeaderWords) * sizeof(u32), s));      // tickets + status words
            goff = q.totals + (size_t)((shift - first_shift) / 8) * 256;
            grid = sweep_grid;
        } else {
            if (first_half_hist && shift == first_shift)
                hipLaunchKernelGGL(k_hist_merge<TILE>, dim3(n_tiles), dim3(256), 0, s, first_half_hist, (n + kBuildBlock - 1) / kBuildBlock, q.hist);
            else {
                hipLaunchKernelGGL(k_radix_hist<TILE>, dim3(n_tiles), dim3(256), 0, s, key[in], n, shift, q.hist);
                c->stats.n_key_hist_launches++;
            }
            hipLaunchKernelGGL(k_radix_chunk_sums, dim3(n_chunks), dim3(256), 0, s, q.hist, n_tiles, q.chunk);
            hipLaunchKernelGGL(k_radix_scan_chunks, dim3(1), dim3(256), 0, s, q.chunk, n_chunks);
            hipLaunchKernelGGL(k_radix_apply, dim3(n_chunks), dim3(256), 0, s, q.hist, n_tiles, q.chunk);
        }
        const bool timing = c->ev_used + 2 <= c->ev_scatter.size();
        if (timing) HIP_TRY(hipEventRecord(c->ev_scatter[c->ev_used], s));
        // the q.hist buffer holds the tile offsets (histogram passes) or the tickets + status words (look-back)
#define MGX_SCATTER(P64, P32, LOW, K_IN, K_OUT, P64_IN, P64_OUT, P32_IN, P32_OUT)                                                                      \
        do {                                                                                                                                           \
            if (sweep) hipLaunchKernelGGL((k_radix_scatter<P64, P32, WAVES, LOW, true>), dim3(grid), dim3(WAVES * 64), 0, s, K_IN, K_OUT, P64_IN, P64_OUT, \
                                          P32_IN, P32_OUT, n, shift, goff, n_tiles, sweep_lag ? 2 : 1, q.hist, sweep_err);                                     \
            else hipLaunchKernelGGL((k_radix_scatter<P64, P32, WAVES, LOW, false>), dim3(grid), dim3(WAVES * 64), 0, s, K_IN, K_OUT, P64_IN, P64_OUT,      \
                                    P32_IN, P32_OUT, n, shift, goff, n_tiles, c->xcd_order, (u32*)nullptr, (u32*)nullptr);                               \
        } while (0)
        if (p64 && !p32)
            MGX_SCATTER(true, false, false, key[in], key[out], p64[in], p64[out], (const u32*)nullptr, (u32*)nullptr);
        else if (p64)
            MGX_SCATTER(true, true, false, key[in], key[out], p64[in], p64[out], p32[in], p32[out]);
        else if (p32)
            MGX_SCATTER(false, true, false, key[in], key[out], (const u64*)nullptr, (u64*)nullptr, p32[in], p32[out]);
        else if (low32_out && shift + 8 >= first_shift + bits) {
            MGX_SCATTER(false, false, true, key[in], key[out], (const u64*)nullptr, (u64*)nullptr, (const u32*)nullptr, low32_out);
            if (low32_done) *low32_done = true;
        } else
            MGX_SCATTER(false, false, false, key[in], key[out], (const u64*)nullptr, (u64*)nullptr, (const u32*)nullptr, (u32*)nullptr);
#undef MGX_SCATTER
        if (timing) { HIP_TRY(hipEventRecord(c->ev_scatter[c->ev_used + 1], s)); c->ev_used += 2; }
        c->scatter_bytes += (uint64_t)n * 2 * (8 + (p32 ? 4 : 0) + (p64 ? 8 : 0));
        if (low32_done && *low32_done) c->scatter_bytes -= (uint64_t)n * 4;
        c->stats.n_radix_passes++;
        *cur = out;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// one stable LSD radix sort of (key, [p64], p32) over `bits` low bits of the key
// buffers are ping-pong pairs; *cur is the index of the input buffer and is updated
int radix_sort(mgx_sortdedup* c, hipStream_t s, const mgx_sortdedup::Scratch& q, u64* key[2], u64* p64[2], u32* p32[2], u32 n,
               int first_shift, int bits, int* cur, u32* low32_out = nullptr, bool* low32_done = nullptr, const u32* first_half_hist = nullptr) {
    // 8192-key tiles (MGX_SORTDEDUP_WIDE_TILES=1) were meant to double the payload run a digit leaves the tile with;
    // measured on an MI355X at 200 M records they LOSE: whole pipeline 10.73 ms against 10.15 ms with 4096-key tiles
    // (64 KB of LDS per workgroup leaves two workgroups per CU), so the default stays 4096
    const bool wide = c->wide_tiles > 0;
    return wide ? radix_sort_w<8>(c, s, q, key, p64, p32, n, first_shift, bits, cur, low32_out, low32_done, first_half_hist)
                : radix_sort_w<kScatterWaves>(c, s, q, key, p64, p32, n, first_shift, bits, cur, low32_out, low32_done, first_half_hist);
}

// duplicate search over one sorted entry array: run heads -> dense list -> marks (+ long runs)
template <bool DOUBLE, bool PK, int KS = 0>
void launch_find(mgx_sortdedup* c, hipStream_t s, const u64* k1, const u64* k2, const u32* rec, u32 n_entries,
                 const mgx_sortdedup::Scratch& q, u32* n_multi, u64 ind_bits, u64 ind_off, u32* sub_start = nullptr, u32 n_sub = 0) {
    if (!n_entries) return;
    const u32 per = 256 * kFindItems;
    hipLaunchKernelGGL((k_find_runs<DOUBLE, PK, KS>), dim3((n_entries + per - 1) / per), dim3(256), 0, s, k1, k2, rec, n_entries,
                       c->d_indicator, ind_bits, ind_off, c->d_dup, q.multi, n_multi, sub_start, n_sub);
}

template <bool DOUBLE, bool PK, int KS = 0>
void launch_mark(mgx_sortdedup* c, hipStream_t s, const u64* k1, const u64* k2, const u32* rec, u32 n_entries,
                 const mgx_sortdedup::Scratch& q, u32* n_multi, u32* n_long, u64 ind_bits, u64 ind_off, bool find_done = false,
                 u32 rec_mask = 0xFFFFFFFFu) {
    if (!n_entries) return;
    if (!find_done) launch_find<DOUBLE, PK, KS>(c, s, k1, k2, rec, n_entries, q, n_multi, ind_bits, ind_off);
    hipLaunchKernelGGL((k_mark_list<DOUBLE, PK, KS>), dim3(c->n_cu * 16), dim3(256), 0, s, k1, k2, rec, n_entries, c->d_recs, c->n,
                       c->d_indicator, ind_bits, ind_off, c->d_dup, q.multi, n_multi, q.longl, n_long, rec_mask);
    if constexpr (KS > kNearScoreBits)
        hipLaunchKernelGGL((k_mark_long_sub<KS>), dim3(c->n_cu * 2), dim3(256), 0, s, k1, rec, n_entries, c->d_recs, c->d_dup, q.longl, n_long,
                           &c->d_sc->huge_runs, rec_mask);
    else
        hipLaunchKernelGGL((k_mark_long<DOUBLE, PK, KS>), dim3(c->n_cu * 2), dim3(256), 0, s, k1, k2, rec, n_entries, c->d_recs,
                           c->d_indicator, ind_bits, ind_off, c->d_dup, q.longl, n_long, rec_mask);
}

}  // namespace

extern "C" {

int mgx_sortdedup_create(int device, unsigned flags, mgx_sortdedup_t** out) {
    if (!out) { set_error("out is NULL"); return -EINVAL; }
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        set_error("no HIP device is visible (this library has no CPU fallback)");
        return -ENODEV;
    }
    if (device == -1) {                 // MGX_DEVICE_AUTO: contexts are dealt round-robin over the visible GPUs
        static std::atomic<unsigned> next{0};
        device = (int)(next.fetch_add(1) % (unsigned)n_dev);
    }
    if (device < 0 || device >= n_dev) { set_error("device %d out of range", device); return -EINVAL; }
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<mgx_sortdedup> c(new (std::nothrow) mgx_sortdedup);
    if (!c) return -ENOMEM;
    c->device = device; c->flags = flags;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount;
    // MGX_STREAM_HIGH_PRIORITY (bit 16): the pipeline's short HBM-bound kernels go ahead of a co-resident PairHMM context's
    int prio_least = 0, prio_greatest = 0;
    const bool high = (flags & (1u << 16)) != 0;
    if (high) HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    auto make_stream = [&](hipStream_t* st) {
        return high ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_greatest) : hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    };
    {
        // optional CU partition (flags bits 8..15: an 8-bit pattern repeated over the CU index): lets a
        // VALU-bound PairHMM context and an HBM-bound sort context share one GPU side by side instead
        // of queueing behind each other (BASELINE.json configs[4])
        const unsigned pat = (flags >> 8) & 0xFFu;
        if (pat != 0 && pat != 0xFFu) {
            const int n_words = (c->n_cu + 31) / 32;
            std::vector<uint32_t> mask(n_words, 0);
            for (int cu = 0; cu < c->n_cu; ++cu) if ((pat >> (cu & 7)) & 1u) mask[cu >> 5] |= 1u << (cu & 31);
            HIP_TRY(hipExtStreamCreateWithCUMask(&c->compute, (uint32_t)n_words, mask.data()));
        } else {
            HIP_TRY(make_stream(&c->compute));
        }
    }
    HIP_TRY(make_stream(&c->copy));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(make_stream(&c->side[i]));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_side[i], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&c->ev_ind, hipEventDisableTiming));
    HIP_TRY(hipMalloc((void**)&c->d_sc, sizeof(Scalars)));
    HIP_TRY(hipEventCreate(&c->ev_start)); HIP_TRY(hipEventCreate(&c->ev_stop));
    c->ev_scatter.resize(64);
    for (auto& e : c->ev_scatter) HIP_TRY(hipEventCreate(&e));
    for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreate(&c->pin_ev[i]));   // staging buffers: on first upload
    *out = c.release();
    return 0;
}

void mgx_sortdedup_destroy(mgx_sortdedup_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    free_buffers(c);
    (void)hipFree(c->d_recs); (void)hipFree(c->d_wire[0]); (void)hipFree(c->d_wire[1]);
    (void)hipFree(c->d_indicator); (void)hipFree(c->d_sub_start); (void)hipFree(c->d_sc);
    (void)hipFree(c->d_ocoord); (void)hipFree(c->d_oarr); (void)hipFree(c->d_marks);
    for (int i = 0; i < 2; ++i) { if (c->pinned[i]) (void)hipHostFree(c->pinned[i]); if (c->pin_ev[i]) (void)hipEventDestroy(c->pin_ev[i]); }
    for (auto e : c->ev_scatter) (void)hipEventDestroy(e);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    for (int i = 0; i < 2; ++i) { if (c->side[i]) (void)hipStreamDestroy(c->side[i]); if (c->ev_side[i]) (void)hipEventDestroy(c->ev_side[i]); }
    if (c->ev_ind) (void)hipEventDestroy(c->ev_ind);
    if (c->compute) (void)hipStreamDestroy(c->compute);
    if (c->copy) (void)hipStreamDestroy(c->copy);
    delete c;
}

namespace {

// device room for `n` records; the first `keep` records already uploaded survive a growth
int ensure_recs(mgx_sortdedup_t* c, size_t n, size_t keep) {
    if (n <= c->recs_cap) return 0;
    size_t cap = std::max<size_t>(n, keep ? c->recs_cap + c->recs_cap / 2 : 0);
    mgx_rec_t* fresh = nullptr;
    if (hipMalloc((void**)&fresh, std::max<size_t>(cap, 1) * sizeof(mgx_rec_t)) != hipSuccess) {
        set_error("out of device memory for %zu records", cap);
        return -ENOMEM;
    }
    if (keep && c->d_recs) {
        HIP_TRY(hipMemcpyAsync(fresh, c->d_recs, keep * sizeof(mgx_rec_t), hipMemcpyDeviceToDevice, c->copy));
        HIP_TRY(hipStreamSynchronize(c->copy));
    }
    (void)hipFree(c->d_recs);
    c->d_recs = fresh; c->recs_cap = cap;
    return 0;
}

// bitmap, sub-tile table, work buffers and the per-input decisions shared by all upload forms
int prepare_input(mgx_sortdedup_t* c, uint64_t L, size_t capacity) {
    int rc = ensure_capacity(c, capacity);
    if (rc) { set_error("out of device memory for %zu records", capacity); return rc; }
    // double_pair_indicator: 4L bits like the reference (main.cpp:115)
    const uint64_t Lp = (L + kIndTile - 1) / kIndTile * kIndTile;          // tile-aligned reverse offset
    const uint64_t bits = std::max<uint64_t>(4 * L + 64, 2 * Lp + 2 * (uint64_t)kIndTile);
    const size_t words = (size_t)((bits + 31) / 32);
    if (words > c->indicator_cap_words) {
        (void)hipFree(c->d_indicator); c->d_indicator = nullptr; c->indicator_cap_words = 0;
        HIP_TRY(hipMalloc((void**)&c->d_indicator, words * 4));
        c->indicator_cap_words = words;
    }
    {
        const size_t n_sub = (size_t)(Lp / kNearSpan) + 2;
        if (n_sub > c->sub_cap) {
            (void)hipFree(c->d_sub_start); c->d_sub_start = nullptr; c->sub_cap = 0;
            HIP_TRY(hipMalloc((void**)&c->d_sub_start, n_sub * 4));
            c->sub_cap = n_sub;
        }
    }
    c->indicator_bits = 4 * L;
    c->L = L; c->ran = false;
    c->near_by_position = true;             // decided again for every input (see finish_run)
    c->packed_coord = L < 0xFFFFFFFFull;     // coord <= L (bam_record.cpp:18-24)
    c->packed_pair = L < 0xF0000000ull;      // 5' ends <= L + clip; verified against the device maximum
    return 0;
}

int ensure_staging(mgx_sortdedup_t* c, size_t chunk) {
    if (chunk <= c->pinned_cap) return 0;
    HIP_TRY(hipStreamSynchronize(c->copy));
    for (int i = 0; i < 2; ++i) {
        if (c->pinned[i]) (void)hipHostFree(c->pinned[i]);
        (void)hipFree(c->d_wire[i]);
        c->pinned[i] = nullptr; c->d_wire[i] = nullptr;
    }
    c->pinned_cap = 0;
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipHostMalloc(&c->pinned[i], chunk, hipHostMallocDefault));
        HIP_TRY(hipMalloc(&c->d_wire[i], chunk));
    }
    c->pinned_cap = chunk;
    return 0;
}

int upload_threads() {
    // one core moves ~12 GB/s into the staging buffer, less than a quarter of what the link takes: every piece is
    // split over a gang of threads (MGX_UPLOAD_THREADS, default 4).  Measured on the GPU box, 200 M records, 24-byte
    // wire form (profiles/r02_upload_threads.txt): 4 threads 89.6 ms, 8: 95.9, 12: 97.4, 16: 113.4 -- beyond four the
    // threads only compete for memory bandwidth; raw 32-byte records with 8 threads: 114.9 ms.
    static const int n_thr = [] {
        const char* e = getenv("MGX_UPLOAD_THREADS");
        const int v = e ? atoi(e) : 4;
        return v < 1 ? 1 : (v > 16 ? 16 : v);
    }();
    return n_thr;
}

extern "C++" {
// A gang of helper threads that lives for one upload: every staging piece (~1 ms of copying) is split over the
// gang without creating threads per piece (eight std::thread starts cost a quarter of a piece's copy time).
class Gang {
public:
    explicit Gang(int n) : n_(n < 1 ? 1 : n) {
        for (int t = 1; t < n_; ++t) th_.emplace_back([this, t] { loop(t); });
    }
    ~Gang() {
        stop_.store(true);
        gen_.fetch_add(1);
        for (auto& t : th_) t.join();
    }
    // body(first, last) over [0, n_items) in n contiguous shares; returns when all shares are done
    template <class F>
    void run(size_t n_items, size_t min_per_thread, F body) {
        if (n_ == 1 || n_items < 2 * min_per_thread) { body((size_t)0, n_items); return; }
        const size_t part = (n_items + (size_t)n_ - 1) / (size_t)n_;
        fn_ = [&, part, n_items](int t) { const size_t a = std::min(n_items, (size_t)t * part), b = std::min(n_items, a + part); if (a < b) body(a, b); };
        done_.store(0);
        gen_.fetch_add(1);
        fn_(0);
        while (done_.load() != n_ - 1) std::this_thread::yield();
    }
private:
    void loop(int t) {
        uint64_t seen = 0;
        for (;;) {
            while (gen_.load() == seen) std::this_thread::yield();
            seen = gen_.load();
            if (stop_.load()) return;
            fn_(t);
            done_.fetch_add(1);
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::function<void(int)> fn_;
    std::atomic<uint64_t> gen_{0};
    std::atomic<int> done_{0};
    std::atomic<bool> stop_{false};
};
}  // extern "C++"

// host memory -> HBM through two pinned staging buffers on the copy stream (asynchronous; the caller syncs)
int stream_up(mgx_sortdedup_t* c, void* dst, const void* src_, size_t total) {
    if (!total) return 0;
    int rc = ensure_staging(c, std::min(kPinnedChunk, std::max<size_t>((total + 1) / 2, 1u << 20)));
    if (rc) return rc;
    size_t off = 0;
    int buf = 0;
    Gang gang(total >= (32u << 20) ? upload_threads() : 1);
    while (off < total) {
        const size_t len = std::min(c->pinned_cap, total - off);
        HIP_TRY(hipEventSynchronize(c->pin_ev[buf]));
        const char* src = reinterpret_cast<const char*>(src_) + off;
        char* stage = static_cast<char*>(c->pinned[buf]);
        gang.run(len, 4u << 20, [=](size_t a, size_t b) { memcpy(stage + a, src + a, b - a); });
        HIP_TRY(hipMemcpyAsync(reinterpret_cast<char*>(dst) + off, c->pinned[buf], len, hipMemcpyHostToDevice, c->copy));
        HIP_TRY(hipEventRecord(c->pin_ev[buf], c->copy));
        off += len; buf ^= 1;
    }
    return 0;
}

// records -> d_recs[first ...): 24-byte wire form whenever the piece allows it (MGX_SORTDEDUP_WIRE=32 keeps the raw form)
int stream_up_recs(mgx_sortdedup_t* c, uint64_t first, uint64_t n, const mgx_rec_t* recs) {
    if (!n) return 0;
    static const bool compact_ok = [] { const char* e = getenv("MGX_SORTDEDUP_WIRE"); return !(e && atoi(e) == 32); }();
    const size_t total = (size_t)n * sizeof(mgx_rec_t);
    int rc = ensure_staging(c, std::min(kPinnedChunk, std::max<size_t>((total + 1) / 2, 1u << 20)));
    if (rc) return rc;
    const size_t piece = c->pinned_cap / sizeof(mgx_rec_t);
    int buf = 0;
    Gang gang(total >= (32u << 20) ? upload_threads() : 1);
    for (uint64_t off = 0; off < n; off += piece, buf ^= 1) {
        const size_t m = (size_t)std::min<uint64_t>(piece, n - off);
        HIP_TRY(hipEventSynchronize(c->pin_ev[buf]));
        const mgx_rec_t* src = recs + off;
        mgx_rec_t* dst = c->d_recs + first + off;
        bool compact = compact_ok;
        if (compact) {
            // a wire record is the low halves of the two 64-bit fields followed by the record's second 16 bytes as
            // they are: four 8-byte loads and three 8-byte stores per record, as cheap as the plain copy
            uint64_t* w = static_cast<uint64_t*>(c->pinned[buf]);
            const uint64_t* s64 = reinterpret_cast<const uint64_t*>(src);
            static_assert(sizeof(mgx_rec_t) == 32 && offsetof(mgx_rec_t, prime5) == 8 && offsetof(mgx_rec_t, mate) == 16, "record layout");
            std::atomic<uint64_t> high{0};
            gang.run(m, 65536, [&, w, s64](size_t a, size_t b) {
                uint64_t hi = 0;
                for (size_t i = a; i < b; ++i) {
                    const uint64_t c0 = s64[4 * i], p0 = s64[4 * i + 1];
                    hi |= c0 | p0;
                    w[3 * i] = (c0 & 0xFFFFFFFFull) | (p0 << 32);
                    w[3 * i + 1] = s64[4 * i + 2];
                    w[3 * i + 2] = s64[4 * i + 3];
                }
                if (hi >> 32) high.fetch_or(1);
            });
            compact = high.load() == 0;
        }
        if (compact) {
            HIP_TRY(hipMemcpyAsync(c->d_wire[buf], c->pinned[buf], m * sizeof(Wire24), hipMemcpyHostToDevice, c->copy));
            hipLaunchKernelGGL(k_expand24, dim3((u32)((m + 255) / 256)), dim3(256), 0, c->copy, (const Wire24*)c->d_wire[buf], dst, (u32)m);
        } else {
            char* stage = static_cast<char*>(c->pinned[buf]);
            const char* s8 = reinterpret_cast<const char*>(src);
            gang.run(m * sizeof(mgx_rec_t), 4u << 20, [=](size_t a, size_t b) { memcpy(stage + a, s8 + a, b - a); });
            HIP_TRY(hipMemcpyAsync(dst, c->pinned[buf], m * sizeof(mgx_rec_t), hipMemcpyHostToDevice, c->copy));
        }
        HIP_TRY(hipEventRecord(c->pin_ev[buf], c->copy));
    }
    return 0;
}

}  // namespace

int mgx_sortdedup_upload_begin(mgx_sortdedup_t* c, uint64_t L, uint64_t n_expected) {
    if (!c) { set_error("ctx is NULL"); return -EINVAL; }
    if (n_expected >= 0xFFFFFFF0ull) { set_error("more than 2^32 records in one shard"); return -E2BIG; }
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_recs(c, (size_t)n_expected, 0);
    if (rc) return rc;
    c->up_L = L; c->up_high = 0; c->uploading = true; c->ran = false;
    return 0;
}

int mgx_sortdedup_upload_chunk(mgx_sortdedup_t* c, uint64_t first_record, uint64_t n_records, const mgx_rec_t* recs) {
    if (!c) { set_error("ctx is NULL"); return -EINVAL; }
    if (!c->uploading) { set_error("mgx_sortdedup_upload_begin has not been called"); return -EINVAL; }
    if (n_records && !recs) { set_error("recs is NULL"); return -EINVAL; }
    if (first_record + n_records >= 0xFFFFFFF0ull) { set_error("more than 2^32 records in one shard"); return -E2BIG; }
    if (first_record > c->up_high) {           // a gap would leave records of the device array undefined
        set_error("chunk starts at record %llu but only %llu records have been uploaded", (unsigned long long)first_record, (unsigned long long)c->up_high);
        return -EINVAL;
    }
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_recs(c, (size_t)(first_record + n_records), (size_t)c->up_high);
    if (rc) return rc;
    if ((rc = stream_up_recs(c, first_record, n_records, recs))) return rc;
    c->up_high = std::max(c->up_high, first_record + n_records);
    return 0;
}

int mgx_sortdedup_upload_end(mgx_sortdedup_t* c, uint64_t n_records) {
    if (!c) { set_error("ctx is NULL"); return -EINVAL; }
    if (!c->uploading) { set_error("mgx_sortdedup_upload_begin has not been called"); return -EINVAL; }
    if (n_records != c->up_high) { set_error("%llu records announced, chunks covered %llu", (unsigned long long)n_records, (unsigned long long)c->up_high); return -EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    c->uploading = false;
    int rc = prepare_input(c, c->up_L, (size_t)n_records);
    if (rc) return rc;
    c->n = c->n_order = (u32)n_records; c->sharded = false; c->n_marks = 0;
    HIP_TRY(hipStreamSynchronize(c->copy));
    return 0;
}

int mgx_sortdedup_upload(mgx_sortdedup_t* c, uint64_t L, uint64_t n_records, const mgx_rec_t* recs) {
    int rc = mgx_sortdedup_upload_begin(c, L, n_records);
    if (!rc) rc = mgx_sortdedup_upload_chunk(c, 0, n_records, recs);
    if (!rc) rc = mgx_sortdedup_upload_end(c, n_records);
    return rc;
}

int mgx_sortdedup_upload_shard(mgx_sortdedup_t* c, uint64_t L, const mgx_sortdedup_shard_t* sh) {
    if (!c || !sh) { set_error("NULL argument"); return -EINVAL; }
    if (sh->n_order >= 0xFFFFFFF0ull || sh->n_mark >= 0xFFFFFFF0ull || sh->n_marks >= 0xFFFFFFF0ull) { set_error("more than 2^32 entries in one shard"); return -E2BIG; }
    if ((sh->n_order && (!sh->order_coord || !sh->order_arrival)) || (sh->n_mark && (!sh->mark_recs || !sh->mark_arrival)) ||
        (sh->n_marks && !sh->marks)) { set_error("a shard array is NULL (was the shard materialised by mgx_sortdedup_route?)"); return -EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    c->uploading = false;
    int rc = ensure_recs(c, (size_t)sh->n_mark, 0);
    if (!rc) rc = prepare_input(c, L, (size_t)std::max(sh->n_order, sh->n_mark));
    if (rc) return rc;
    if (sh->n_order > c->order_cap) {
        (void)hipFree(c->d_ocoord); (void)hipFree(c->d_oarr); c->d_ocoord = nullptr; c->d_oarr = nullptr; c->order_cap = 0;
        HIP_TRY(hipMalloc((void**)&c->d_ocoord, sh->n_order * 8)); HIP_TRY(hipMalloc((void**)&c->d_oarr, sh->n_order * 4));
        c->order_cap = sh->n_order;
    }
    if (sh->n_marks > c->marks_cap) {
        (void)hipFree(c->d_marks); c->d_marks = nullptr; c->marks_cap = 0;
        HIP_TRY(hipMalloc((void**)&c->d_marks, sh->n_marks * 8));
        c->marks_cap = sh->n_marks;
    }
    c->n = (u32)sh->n_mark; c->n_order = (u32)sh->n_order; c->n_marks = (u32)sh->n_marks; c->sharded = true;
    if ((rc = stream_up_recs(c, 0, sh->n_mark, sh->mark_recs))) return rc;
    if ((rc = stream_up(c, c->d_ocoord, sh->order_coord, (size_t)sh->n_order * 8))) return rc;
    if ((rc = stream_up(c, c->d_oarr, sh->order_arrival, (size_t)sh->n_order * 4))) return rc;
    if ((rc = stream_up(c, c->d_marks, sh->marks, (size_t)sh->n_marks * 8))) return rc;
    HIP_TRY(hipStreamSynchronize(c->copy));
    return 0;
}

int mgx_sortdedup_run(mgx_sortdedup_t* c) {
    if (!c) { set_error("ctx is NULL"); return -EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->compute;
    const u32 n = c->n, n_order = c->n_order;
    c->stats = mgx_sortdedup_stats_t{};
    c->stats.n_records = n_order;
    c->ev_used = 0; c->scatter_bytes = 0;
    HIP_TRY(hipEventRecord(c->ev_start, s));
    for (int attempt = 0; attempt < 2; ++attempt) {
        HIP_TRY(hipMemsetAsync(c->d_sc, 0, sizeof(Scalars), s));
        if (n) {
            HIP_TRY(hipMemsetAsync(c->d_dup, 0, n, s));
            const u32 nb = (n + kBuildBlock - 1) / kBuildBlock;
            BuildOut o{c->sharded ? nullptr : c->d_ckey[0], c->d_cval[0], c->d_k1[0], c->d_k2[0], c->d_prec[0], c->d_sk1[0], c->d_srec[0],
                       c->d_indicator, c->indicator_bits, c->L, c->packed_coord ? 1 : 0, c->packed_pair ? 1 : 0,
                       c->d_nk[0], c->d_nrec[0], c->packed_pair ? 1 : 0, n < 0x80000000u ? 0x80000000u : 0u,
                       c->sharded ? nullptr : c->d_half_hist};
            hipLaunchKernelGGL(k_build_emit, dim3(nb), dim3(256), 0, s, c->d_recs, n, o, c->d_sc);
            HIP_TRY(hipGetLastError());
        }
        if (c->sharded && n_order)
            hipLaunchKernelGGL(k_order_keys, dim3(std::min<u32>((n_order + 255) / 256, (u32)c->n_cu * 16)), dim3(256), 0, s, c->d_ocoord, c->d_oarr,
                               n_order, c->d_ckey[0], c->d_cval[0], c->packed_coord ? 1 : 0, c->d_sc);
        // the only host round trip: entry counts and key maxima size the sorts
        HIP_TRY(hipMemcpyAsync(&c->sc, c->d_sc, sizeof(Scalars), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        // a coordinate beyond L (a position past the end of its contig) does not fit the packed
        // (coord << 32 | index) key: rebuild once with separate key/value arrays
        if ((c->packed_coord && c->sc.max_coord >= (1ull << 32)) || (c->packed_pair && c->sc.max_k2d >= (1ull << 32))) {
            if (c->sc.max_coord >= (1ull << 32)) c->packed_coord = false;
            if (c->sc.max_k2d >= (1ull << 32)) c->packed_pair = false;
            continue;
        }
        break;
    }
    if (c->sc.bad_mate) { set_error("a record's mate index is outside the uploaded shard"); return -EINVAL; }
    const u32 nd = c->sc.n_double, ns = c->sc.n_single, nn = c->sc.n_near;   // nd: far pairs only
    c->stats.n_double = (uint64_t)nd + nn; c->stats.n_single = ns;
    c->stats.key_bits_coord = bits_of(c->sc.max_coord);
    c->stats.key_bits_pair1 = bits_of(std::max<uint64_t>(c->sc.max_k1d, c->sc.max_k1s));
    c->stats.key_bits_pair2 = bits_of(c->sc.max_k2d);
    int rc;
    // indicator bitmap: tiled (plain stores, reverse half at a tile-aligned offset) whenever every 5' end
    // is safely below L; otherwise the reference's exact layout with global atomics
    const uint64_t maxpos = std::max<uint64_t>(c->sc.max_k2d, std::max<uint64_t>(c->sc.max_k1d >> 2, c->sc.max_k1s >> 2));
    const bool tiled = c->L > 64 && maxpos < c->L - 64;
    const uint64_t Lp = (c->L + kIndTile - 1) / kIndTile * kIndTile;
    const uint64_t ind_off = tiled ? Lp : c->L;                 // offset of the reverse-strand half
    const uint64_t ind_bits = tiled ? 2 * Lp : c->indicator_bits;
    const u32 n_ind_tiles = (u32)((c->L + kIndTile - 1) / kIndTile);
    if (!tiled && n) HIP_TRY(hipMemsetAsync(c->d_indicator, 0, (size_t)((c->indicator_bits + 64 + 31) / 32) * 4, s));

    // near double pairs (mate 5' end within 65 535 of record 1's -- every proper pair): the whole
    // (sort_key, mate end) identity is ONE injective key word p1 << 16 | orient << 14 | delta, so one
    // LSD sort of (8-byte key, 4-byte record) groups equal pairs: 7 passes x 12 B instead of 9 x 16 B.
    // Runs only need equal keys to be adjacent, not the reference's exact order.
    // The three sorts are independent: near pairs go to side stream 0, records to side stream 1, far
    // pairs and singles stay on the main stream, so their short histogram / scan launches overlap the
    // other sorts' bandwidth-bound scatters.  In-process A/B on one device at 200 M records
    // (tools/dev_sort_ab.py): 18.0 ms against 19.4 ms on a single stream (MGX_SORTDEDUP_STREAMS=1).
    if (const char* e = getenv("MGX_SORTDEDUP_XCD_ORDER")) c->xcd_order = atoi(e) != 0;
    // the look-back scatters are OPT-IN: bit-identical, but measured slower than histogram + scan + scatter on this part
    // (record sort at 200 M keys: 1.20-1.32 ms per pass against 0.36 + 0.06 + 0.68; DESIGN.md 4.2, profiles/r03_sort_onesweep_ab.txt)
    { const char* e = getenv("MGX_SORTDEDUP_ONESWEEP"); c->onesweep = !c->onesweep_failed && (e ? atoi(e) != 0 : false); }
    for (auto& q : c->scr) if (q.totals) HIP_TRY(hipMemsetAsync(q.totals + kMaxPasses * 256, 0, 64 * sizeof(u32), s));      // the look-back error words
    { const char* e = getenv("MGX_SORTDEDUP_WIDE_TILES"); c->wide_tiles = e ? (atoi(e) != 0) : -1; }
    if (const char* e = getenv("MGX_SORTDEDUP_NEAR_EXACT")) { if (atoi(e) != 0) c->near_by_position = false; }
    const char* env_streams = getenv("MGX_SORTDEDUP_STREAMS");
    const bool multi = !(env_streams && atoi(env_streams) == 1);
    hipStream_t sN = (multi && tiled) ? c->side[0] : s;   // the atomic (non-tiled) bitmap needs the memset first
    hipStream_t sR = multi ? c->side[1] : s;
    int ncur = 0;
    const int near_shift = c->near_by_position ? kNearShift : kNearScoreBits;
    if ((rc = radix_sort(c, sN, c->scr[1], c->d_nk, nullptr, c->d_nrec, nn, near_shift,
                         std::max(bits_of(c->sc.max_near) - near_shift, 1), &ncur))) return rc;
    bool defined = false;                     // has a pass already written every word of the tiled bitmap?
    const bool near_tiles = tiled && n && c->packed_pair;
    const u32 n_sub = (u32)(Lp / kNearSpan);
    // run heads first: the same pass over the sorted keys notes where they cross the bitmap's sub-tile boundaries
    if (near_tiles && !nn) HIP_TRY(hipMemsetAsync(c->d_sub_start, 0, ((size_t)n_sub + 1) * 4, sN));
    if (c->near_by_position)
        launch_find<true, false, kNearShift>(c, sN, c->d_nk[ncur], nullptr, c->d_nrec[ncur], nn, c->scr[1], &c->d_sc->n_multi_n, ind_bits, ind_off,
                                             near_tiles ? c->d_sub_start : nullptr, n_sub);
    else
        launch_find<true, false, kNearScoreBits>(c, sN, c->d_nk[ncur], nullptr, c->d_nrec[ncur], nn, c->scr[1], &c->d_sc->n_multi_n, ind_bits, ind_off,
                                                 near_tiles ? c->d_sub_start : nullptr, n_sub);
    if (near_tiles) {
        hipLaunchKernelGGL(k_indicator_tiles_near, dim3(n_ind_tiles), dim3(256), 0, sN, c->d_nk[ncur], nn, c->d_indicator, Lp, c->d_sub_start, n_sub);
        defined = true;
    } else if (!tiled && nn) {
        hipLaunchKernelGGL(k_set_indicator_near, dim3((nn + 255) / 256), dim3(256), 0, sN, c->d_nk[ncur], nn, c->d_indicator, c->indicator_bits, c->L);
    }
    HIP_TRY(hipEventRecord(c->ev_ind, sN));
    const u32 near_rec_mask = n < 0x80000000u ? 0x7FFFFFFFu : 0xFFFFFFFFu;
    if (c->near_by_position)
        launch_mark<true, false, kNearShift>(c, sN, c->d_nk[ncur], nullptr, c->d_nrec[ncur], nn, c->scr[1], &c->d_sc->n_multi_n, &c->d_sc->n_long_n, ind_bits, ind_off, true, near_rec_mask);
    else
        launch_mark<true, false, kNearScoreBits>(c, sN, c->d_nk[ncur], nullptr, c->d_nrec[ncur], nn, c->scr[1], &c->d_sc->n_multi_n, &c->d_sc->n_long_n, ind_bits, ind_off, true, near_rec_mask);
    HIP_TRY(hipEventRecord(c->ev_side[0], sN));

    // records by unified coordinate (stable: equal coordinates keep arrival order)
    int ccur = 0;
    bool unpacked = false;                    // did the last pass store the order directly?
    c->ev_rec_begin = c->ev_used;
    const uint64_t bytes_before_records = c->scatter_bytes;
    if (c->packed_coord) {
        // key = coord << 32 | arrival index: sort on the high half only, 8 bytes per record per pass
        if ((rc = radix_sort(c, sR, c->scr[2], c->d_ckey, nullptr, nullptr, n_order, 32, bits_of(c->sc.max_coord), &ccur, c->d_cval[0], &unpacked,
                             c->sharded ? nullptr : c->d_half_hist))) return rc;
        if (n_order && !unpacked) hipLaunchKernelGGL(k_unpack_order, dim3((n_order + 255) / 256), dim3(256), 0, sR, c->d_ckey[ccur], n_order, c->d_cval[0]);
        ccur = 0;
    } else {
        if ((rc = radix_sort(c, sR, c->scr[2], c->d_ckey, nullptr, c->d_cval, n_order, 0, bits_of(c->sc.max_coord), &ccur, nullptr, nullptr,
                             c->sharded ? nullptr : c->d_half_hist))) return rc;
    }
    c->order_buf = ccur;
    c->ev_rec_end = c->ev_used;
    c->rec_scatter_bytes = c->scatter_bytes - bytes_before_records;
    if (unpacked && c->ev_rec_end >= c->ev_rec_begin + 2) {
        // the statistics of "the record-sort scatter kernel" cover the full-width launches only; the
        // last one is a different instantiation (half the store bytes)
        c->ev_rec_end -= 2;
        c->rec_scatter_bytes -= (uint64_t)n_order * 12;
    }
    HIP_TRY(hipEventRecord(c->ev_side[1], sR));

    // far double pairs (discordant, cross-contig, or every pair when keys are wider than 32 bits):
    // LSD over (sort_key, mate 5' end): sort by the mate end first, then by sort_key.
    // Packed form: two 8-byte arrays per entry -- sort_key and (mate end << 32 | record) -- take
    // turns as key and payload (16 B per entry per pass, two LDS exchange rounds instead of three).
    int cur = 0;
    const bool pk = c->packed_pair;
    if (pk) {
        if ((rc = radix_sort(c, s, c->scr[0], c->d_k2, c->d_k1, nullptr, nd, 32, bits_of(c->sc.max_k2d), &cur))) return rc;
    } else {
        if ((rc = radix_sort(c, s, c->scr[0], c->d_k2, c->d_k1, c->d_prec, nd, 0, bits_of(c->sc.max_k2d), &cur))) return rc;
    }
    HIP_TRY(hipStreamWaitEvent(s, c->ev_ind, 0));      // the near pass defines the bitmap's words first
    if (tiled && n && (nd || !defined)) {
        const dim3 gt(n_ind_tiles);
        if (defined) {
            if (pk) hipLaunchKernelGGL((k_indicator_tiles<2, true, false>), gt, dim3(256), 0, s, c->d_k1[cur], c->d_k2[cur], nd, c->d_indicator, Lp);
            else    hipLaunchKernelGGL((k_indicator_tiles<2, false, false>), gt, dim3(256), 0, s, c->d_k1[cur], c->d_k2[cur], nd, c->d_indicator, Lp);
        } else {
            if (pk) hipLaunchKernelGGL((k_indicator_tiles<2, true, true>), gt, dim3(256), 0, s, c->d_k1[cur], c->d_k2[cur], nd, c->d_indicator, Lp);
            else    hipLaunchKernelGGL((k_indicator_tiles<2, false, true>), gt, dim3(256), 0, s, c->d_k1[cur], c->d_k2[cur], nd, c->d_indicator, Lp);
            defined = true;
        }
    } else if (!tiled && nd) {
        if (pk) hipLaunchKernelGGL((k_set_indicator<2, true>), dim3((nd + 255) / 256), dim3(256), 0, s, c->d_k1[cur], c->d_k2[cur], nd, c->d_indicator, c->indicator_bits, c->L);
        else    hipLaunchKernelGGL((k_set_indicator<2, false>), dim3((nd + 255) / 256), dim3(256), 0, s, c->d_k1[cur], c->d_k2[cur], nd, c->d_indicator, c->indicator_bits, c->L);
    }
    if (n && c->n_marks)      // ends of pairs that live in other shards
        hipLaunchKernelGGL(k_or_marks, dim3((c->n_marks + 255) / 256), dim3(256), 0, s, c->d_marks, c->n_marks, c->d_indicator, ind_bits, ind_off,
                           tiled ? c->L - 64 : ~0ull);
    if ((rc = radix_sort(c, s, c->scr[0], c->d_k1, c->d_k2, pk ? nullptr : c->d_prec, nd, 0, bits_of(c->sc.max_k1d), &cur))) return rc;
    if (nd) {
        const dim3 g((nd + 255) / 256);
        if (tiled) {
            if (pk) hipLaunchKernelGGL((k_indicator_tiles<1, true, false>), dim3(n_ind_tiles), dim3(256), 0, s, c->d_k1[cur], c->d_k2[cur], nd, c->d_indicator, Lp);
            else    hipLaunchKernelGGL((k_indicator_tiles<1, false, false>), dim3(n_ind_tiles), dim3(256), 0, s, c->d_k1[cur], c->d_k2[cur], nd, c->d_indicator, Lp);
        } else {
            if (pk) hipLaunchKernelGGL((k_set_indicator<1, true>), g, dim3(256), 0, s, c->d_k1[cur], c->d_k2[cur], nd, c->d_indicator, c->indicator_bits, c->L);
            else    hipLaunchKernelGGL((k_set_indicator<1, false>), g, dim3(256), 0, s, c->d_k1[cur], c->d_k2[cur], nd, c->d_indicator, c->indicator_bits, c->L);
        }
        if (pk) launch_mark<true, true>(c, s, c->d_k1[cur], c->d_k2[cur], nullptr, nd, c->scr[0], &c->d_sc->n_multi_d, &c->d_sc->n_long_d, ind_bits, ind_off);
        else    launch_mark<true, false>(c, s, c->d_k1[cur], c->d_k2[cur], c->d_prec[cur], nd, c->scr[0], &c->d_sc->n_multi_d, &c->d_sc->n_long_d, ind_bits, ind_off);
    }
    // singles
    int scur = 0;
    if ((rc = radix_sort(c, s, c->scr[0], c->d_sk1, nullptr, c->d_srec, ns, 0, bits_of(c->sc.max_k1s), &scur))) return rc;
    launch_mark<false, false>(c, s, c->d_sk1[scur], nullptr, c->d_srec[scur], ns, c->scr[0], &c->d_sc->n_multi_s, &c->d_sc->n_long_s, ind_bits, ind_off);
    HIP_TRY(hipStreamWaitEvent(s, c->ev_side[0], 0));
    HIP_TRY(hipStreamWaitEvent(s, c->ev_side[1], 0));
    if (n) hipLaunchKernelGGL(k_count_dup, dim3(c->n_cu * 4), dim3(256), 0, s, c->d_dup, n, c->d_sc);
    HIP_TRY(hipEventRecord(c->ev_stop, s));
    HIP_TRY(hipGetLastError());
    c->ran = true;
    c->finished = false;
    return 0;
}

// Waits for the pipeline and applies the one fallback that can only be decided afterwards: a position
// holding more near pairs than the in-run comparison accepts sends the input through the pipeline again
// with the six-pass near sort (equal identities adjacent, no quadratic step).
static int finish_run(mgx_sortdedup_t* c) {
    if (c->finished) return 0;
    HIP_TRY(hipStreamSynchronize(c->compute));
    HIP_TRY(hipMemcpy(&c->sc, c->d_sc, sizeof(Scalars), hipMemcpyDeviceToHost));
    u32 sweep_err = 0;
    for (auto& q : c->scr) if (q.totals) { u32 e = 0; HIP_TRY(hipMemcpy(&e, q.totals + kMaxPasses * 256, sizeof e, hipMemcpyDeviceToHost)); sweep_err |= e; }
    if (sweep_err) {
        // a look-back wait ran into its bound (never seen; the protocol's progress argument assumes in-order dispatch, which
        // HIP does not promise): the results of that run are void, run again with the histogram passes and keep to them
        fprintf(stderr, "mgx_sortdedup: a look-back wait timed out; re-running with histogram passes\n");
        c->onesweep_failed = true;
        const int rc = mgx_sortdedup_run(c);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(c->compute));
        HIP_TRY(hipMemcpy(&c->sc, c->d_sc, sizeof(Scalars), hipMemcpyDeviceToHost));
    }
    if (c->sc.huge_runs && c->near_by_position) {
        c->near_by_position = false;
        const int rc = mgx_sortdedup_run(c);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(c->compute));
        HIP_TRY(hipMemcpy(&c->sc, c->d_sc, sizeof(Scalars), hipMemcpyDeviceToHost));
    }
    c->finished = true;
    return 0;
}

int mgx_sortdedup_results(mgx_sortdedup_t* c, uint32_t* out_order, uint8_t* out_dup) {
    if (!c) { set_error("ctx is NULL"); return -EINVAL; }
    if (!c->ran) { set_error("mgx_sortdedup_run has not been called"); return -EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = finish_run(c); if (rc) return rc; }
    if (c->n) {
        if (out_dup) HIP_TRY(hipMemcpyAsync(out_dup, c->d_dup, (size_t)c->n, hipMemcpyDeviceToHost, c->compute));
    }
    if (c->n_order && out_order) HIP_TRY(hipMemcpyAsync(out_order, c->d_cval[c->order_buf], (size_t)c->n_order * 4, hipMemcpyDeviceToHost, c->compute));
    HIP_TRY(hipStreamSynchronize(c->compute));
    return 0;
}

int mgx_sortdedup_stats(mgx_sortdedup_t* c, mgx_sortdedup_stats_t* out) {
    if (!c || !out) { set_error("NULL argument"); return -EINVAL; }
    if (!c->ran) { set_error("mgx_sortdedup_run has not been called"); return -EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = finish_run(c); if (rc) return rc; }
    mgx_sortdedup_stats_t st = c->stats;
    st.n_dup_records = c->sc.n_dup;
    HIP_TRY(hipEventElapsedTime(&st.ms_total, c->ev_start, c->ev_stop));
    st.ms_radix_scatter = 0;
    for (size_t k = 0; k + 1 < c->ev_used + 1 && k + 1 < c->ev_scatter.size() + 1 && k < c->ev_used; k += 2) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_scatter[k], c->ev_scatter[k + 1]));
        st.ms_radix_scatter += ms;
    }
    st.radix_scatter_bytes = c->scatter_bytes;
    st.ms_scatter_records = 0; st.n_scatter_records = 0; st.scatter_records_bytes = c->rec_scatter_bytes;
    for (size_t k = c->ev_rec_begin; k + 1 < c->ev_rec_end + 1 && k < c->ev_rec_end; k += 2) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_scatter[k], c->ev_scatter[k + 1]));
        st.ms_scatter_records += ms; st.n_scatter_records++;
    }
    // LSD-8 traffic model, SURVEY.md section 8d
    const uint64_t N = c->n_order, P = (uint64_t)st.n_double + st.n_single;
    const uint64_t pc = (st.key_bits_coord + 7) / 8, pp = (st.key_bits_pair1 + st.key_bits_pair2 + 7) / 8;
    st.alg_bytes = N * (8 + pc * 2 * 12) + P * (16 + pp * 2 * 20) + P * 29 + N;
    *out = st;
    return 0;
}

int mgx_sortdedup_sort_mark(mgx_sortdedup_t* c, uint64_t L, uint64_t n_records, const mgx_rec_t* recs,
                            uint32_t* out_order, uint8_t* out_dup) {
    int rc = mgx_sortdedup_upload(c, L, n_records, recs);
    if (!rc) rc = mgx_sortdedup_run(c);
    if (!rc) rc = mgx_sortdedup_results(c, out_order, out_dup);
    return rc;
}

}  // extern "C"
