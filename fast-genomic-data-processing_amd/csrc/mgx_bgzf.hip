// mgx_bgzf.hip -- BGZF block compression on gfx950 (C ABI: include/mgx_bgzf.h), SURVEY.md 8f row F3.
//
// What the reference's writer threads do with zlib, one 64 KB block at a time (htslib bgzf_compress,
// deepmutect/htslib/bgzf.c:610-648, from sortmardup/main.cpp:371-421), is done here with one workgroup per
// BGZF block; a batch holds thousands of independent blocks, so the device is filled by blocks, not by the
// inside of one.  Per block, all in LDS (the 64 KB of input, a 4096-entry hash table, the code tables):
//
//   1. match search, positions in order, one window of NT positions per step: a position looks up the
//      hash of its 4 bytes, THEN the window's positions enter the table (atomic max: the nearest earlier
//      occurrence wins, so the result does not depend on scheduling), and extends the candidate -- and the
//      distance-1 candidate, the run-length case of quality strings -- byte-parallel up to 258 bytes;
//   2. parse: every thread walks its own 1/NT of the block (greedy with one-step lazy evaluation, matches cut at
//      the thread's boundary) and emits tokens, counting literal/length and distance symbols in LDS histograms;
//   3. two length-limited Huffman codes per block (rank by counting, a two-queue merge on one lane, frequencies
//      halved and rebuilt in the rare case a code comes out longer than 15 bits), the code-length code and the
//      run-length coded header of RFC 1951 section 3.2.7;
//   4. every thread's tokens are measured, a scan gives its bit offset, and the bits go into the LDS buffer that
//      held the input (OR into 32-bit words); a block that would not shrink is emitted stored;
//   5. CRC-32: per-thread table-driven CRCs of the 1/NT pieces, combined by multiplication with x^(8 * bytes after)
//      modulo the CRC polynomial.
// A finished block (18-byte BGZF header, payload, CRC, ISIZE) lands in a 64 KB slot; a scan over the sizes and a
// copy kernel pack the slots back to back for ONE device-to-host copy.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/mgx_bgzf.h"
#include "mgx_common.h"

using mgx::set_error;

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) { set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return -EIO; } \
    } while (0)

namespace {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

constexpr u32 kWin = 896;                       // positions per match window: 14 wavefronts extend matches, the 15th parses the previous group, the 16th looks up hash candidates
constexpr u32 kParse0 = kWin, kProd0 = kWin + 64; // first thread of the parsing and of the candidate-producing wavefront
constexpr u32 kGrp = 4;                         // match windows per GROUP: 3840 positions = 60 chunks of 64 are matched, parsed and turned into tokens before the next
constexpr int kNT = 1024;                      // threads per workgroup = positions per match window (16 wavefronts: one block per CU, LDS-bound)
constexpr int kHashBits = 12;
constexpr u32 kMaxIn = MGX_BGZF_MAX_BLOCK_IN;
constexpr u32 kSlot = 0x10000;                 // bytes of device scratch per finished block
constexpr u32 kSlotSkew = 2;                   // a block starts at slot + 2: its payload (18 bytes in) is 4-byte aligned
constexpr u32 kPad = 320;                      // zero bytes after the input in LDS (match extension reads ahead)
constexpr u32 kCrcPoly = 0xEDB88320u;
constexpr int kNumLL = 286, kNumD = 30, kNumCL = 19;
constexpr int kAhead = 4;                      // windows whose tokens are fetched ahead of their use (one memory round trip per 16)
constexpr u32 kScratchPerWg = 65536u + kAhead * 1024u;            // u32 per workgroup: the block's packed tokens, densely (the only per-block state that leaves the LDS)
constexpr u32 kGrpPos = kGrp * kWin, kGrpChunks = kGrpPos / 64;
static_assert(kGrpPos % 64 == 0, "a group is a whole number of 64-position chunks");

enum { V_OVER = 0, V_NUSED, V_K, V_CRCLAST, V_NTOK, V_NLIT, V_NDIST, V_NCLSYM, V_HDRBITS, V_STORED, V_CRC, V_CARRY, V_N };

struct __attribute__((aligned(16))) Lds {
    u32 buf[(0x10000 + 512) / 4];              // the block's bytes (at the source's alignment), later the output words
    u32 head[1 << kHashBits];                  // hash -> last position + 1
    u32 crc_tab[4][256];                       // slicing-by-4 tables
    u32 x2n[32];                               // x^(2^k) mod P
    u32 tpow[kNT];                             // x^(512 j) mod P
    u32 xr[65];                                // x^(8 r) mod P
    u32 f_ll[288], f_d[32], f_cl[20];          // symbol counts
    u16 c_ll[288], c_d[32], c_cl[20];          // codes, bit-reversed for LSB-first output
    u8 l_ll[288], l_d[32], l_cl[20];           // code lengths
    u16 sorted[288];
    u16 qid[2][288];                           // Huffman rounds: node ids of the active list (two buffers)
    u32 wl[288], wi[288];                      // weights: leaves (sorted), internal nodes (in creation order)
    u16 parent[576];
    u8 depth[576];
    u32 bl_count[16], next_code[16];
    u32 wsum[kNT / 64];
    u64 smask[6];                              // header: which of the (up to 316) code lengths start a run
    u16 cand[2][kGrp * kWin];                          // hash candidates (position + 1) of the current and the next group
    alignas(16) u32 gm[2][kGrp * kWin];                // the decided matches of the group being matched and of the one before, len << 16 | dist (0: literal)
    u32 cend[64];                              // parse: the index of the first token of chunk l of the group
    u64 mask[64];                              // parse: the positions of chunk l of the group that start a token
    u32 fw[288];                               // huff_build's working copy of the counts
    u8 bcost[256];                             // estimated cost of a literal byte in 1/16 bit, from the block's byte histogram
    u32 hdr[192];                              // the dynamic block header, as bits
    u16 clsym[320];                            // code-length symbols of the header: symbol | extra << 8
    u32 vars[V_N];
};

struct DeflateArgs {
    const u8* in;          // uncompressed bytes of the batch
    const u64* off;        // [n_blocks + 1]
    u32 n_blocks;
    u8* slots;             // [n_blocks][kSlot]
    u32* sizes;            // [n_blocks] bytes of the finished block
    u32* scratch;          // [gridDim.x][kScratchPerWg]: the block's tokens
    u32* n_stored;         // counter
    u32 lazy;
    u32 cost_base, cost_rle;    // estimated bits of a match's length + distance codes (distance 1: cost_rle); 0 = take every match
    unsigned long long* prof;   // optional [8]: cycles per phase, summed over blocks (MGX_BGZF_PROF=1)
};

__device__ __forceinline__ u32 load32(const u8* p) { u32 v; __builtin_memcpy(&v, p, 4); return v; }

__device__ __forceinline__ u32 multmodp(u32 a, u32 b) {
    // product of two polynomials modulo the CRC polynomial, bit 31 = x^0 (reflected)
    u32 p = 0;
#pragma unroll 4
    for (int i = 0; i < 32; ++i) {
        if (a & (0x80000000u >> i)) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}

// length 3..258 -> symbol 257..285, number of extra bits, extra value (RFC 1951 3.2.5)
__device__ __forceinline__ void length_code(u32 length, u32* sym, u32* eb, u32* ev) {
    const u32 l = length - 3;
    if (l < 8) { *sym = 257 + l; *eb = 0; *ev = 0; return; }
    if (l == 255) { *sym = 285; *eb = 0; *ev = 0; return; }
    const u32 msb = 31 - __builtin_clz(l);
    *sym = 257 + 4 * (msb - 1) + ((l >> (msb - 2)) & 3);
    *eb = msb - 2;
    *ev = l & ((1u << (msb - 2)) - 1);
}
// distance 1..32768 -> symbol 0..29
__device__ __forceinline__ void dist_code(u32 dist, u32* sym, u32* eb, u32* ev) {
    const u32 x = dist - 1;
    if (x < 4) { *sym = x; *eb = 0; *ev = 0; return; }
    const u32 msb = 31 - __builtin_clz(x);
    const u32 c = 2 * msb + ((x >> (msb - 1)) & 1);
    *sym = c;
    *eb = (c >> 1) - 1;
    *ev = x & ((1u << *eb) - 1);
}

__device__ __forceinline__ u64 load64(const u8* p) { u64 v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ u32 match_len(const u8* a, const u8* b, u32 maxlen) {
    u32 len = 0;
    while (len + 8 <= maxlen) {
        const u64 x = load64(a + len) ^ load64(b + len);
        if (x) return len + (u32)(__builtin_ctzll(x) >> 3);
        len += 8;
    }
    while (len + 4 <= maxlen) {
        const u32 x = load32(a + len) ^ load32(b + len);
        if (x) return len + (__builtin_ctz(x) >> 3);
        len += 4;
    }
    while (len < maxlen && a[len] == b[len]) ++len;
    return len;
}

// Wavefront-wide inclusive scans on the DPP network (row shifts, then the row broadcasts of gfx9): six VALU operations,
// no trip through the LDS crossbar as __shfl_up takes.  Lane 63 ends up with the reduction over the wavefront.
template <typename Op>
__device__ __forceinline__ u32 wave_scan_inclusive(u32 v, Op op) {
    v = op(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));     // row_shr:1
    v = op(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));     // row_shr:2
    v = op(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));     // row_shr:4
    v = op(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));     // row_shr:8
    v = op(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));     // row_bcast:15 into rows 1 and 3
    v = op(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));     // row_bcast:31 into rows 2 and 3
    return v;
}
struct OpAdd { __device__ __forceinline__ u32 operator()(u32 a, u32 b) const { return a + b; } };
struct OpXor { __device__ __forceinline__ u32 operator()(u32 a, u32 b) const { return a ^ b; } };
__device__ __forceinline__ u32 wave_last(u32 v) { return (u32)__builtin_amdgcn_readlane((int)v, 63); }

// Workgroup-wide exclusive prefix sum (wave shuffles, then the wave totals through LDS).  *total = sum of all values.
__device__ __forceinline__ u32 block_scan(Lds& L, u32 v, u32* total) {
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const u32 inc = wave_scan_inclusive(v, OpAdd());
    __syncthreads();                                   // earlier readers of L.wsum are done
    if (lane == 63) L.wsum[wave] = inc;
    __syncthreads();
    u32 before = 0, all = 0;
#pragma unroll
    for (u32 w = 0; w < kNT / 64; ++w) { const u32 t = L.wsum[w]; all += t; if (w < wave) before += t; }
    *total = all;
    return before + inc - v;
}

// Length-limited Huffman code of freq[0, N): lengths and (bit-reversed) canonical codes.  Called by the whole workgroup.
// At least two symbols get a code (inflate accepts no incomplete literal/length or code-length code).
// Parallel but for the merge itself: ranks by counting, leaf depths by walking up the parent links, code values by
// counting the earlier symbols of the same length.
// WAVE (N <= 64: the distance and the code-length alphabets): wavefront 0 does it alone -- the LDS executes a wavefront's
// instructions in order, so its lanes need no workgroup barrier between the steps -- and the others wait at the exit.
template <bool WAVE>
__device__ void huff_build(Lds& L, const u32* freq_in, const int N, const int limit, u8* len, u16* code, unsigned long long* prof = nullptr) {
    const int tid = (int)threadIdx.x;
    long long tp = prof ? clock64() : 0;
    auto lap = [&](int k) { if (prof && tid == 0) { const long long t = clock64(); atomicAdd(&prof[k], (unsigned long long)(t - tp)); tp = t; } };
    auto sync = [] {
        if constexpr (WAVE) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
        else __syncthreads();
    };
    __syncthreads();
    if (WAVE && tid >= 64) { __syncthreads(); return; }
    if (tid < N) L.fw[tid] = freq_in[tid];            // the caller's counts stay as they are (they price the block)
    u32* const freq = L.fw;
    sync();
    const int used = WAVE ? (int)__popcll(__ballot(tid < N && freq[tid] != 0)) : __syncthreads_count(tid < N && freq[tid] != 0);
    if (tid == 0) {
        if (used == 0) { freq[0] = 1; freq[1] = 1; }
        else if (used == 1) freq[freq[0] ? 1 : 0] = 1;
    }
    lap(8);
    for (;;) {
        sync();
        if (tid < 16) L.bl_count[tid] = 0;
        if (tid == 0) { L.vars[V_NUSED] = 0; L.vars[V_OVER] = 0; }
        sync();
        if (tid < N) {
            len[tid] = 0;
            const u32 f = freq[tid];
            if (f) {
                int r = 0;
                for (int j = 0; j < N; ++j) { const u32 g = freq[j]; r += (g != 0) & ((g < f) | ((g == f) & (j < tid))); }
                L.sorted[r] = (u16)tid;
                L.wl[r] = f;
                atomicAdd(&L.vars[V_NUSED], 1u);
            }
        }
        sync();
        lap(9);
        const int n = (int)L.vars[V_NUSED];
        {
            // Huffman's merges in rounds.  The active nodes form one sorted list Q.  t = Q[0] + Q[1] is the lightest node that
            // can still be created, so every active node of weight <= t is merged before any new node is touched: the
            // (even-sized) prefix of Q up to t pairs up neighbour with neighbour, exactly as the sequential algorithm
            // would take them one pair at a time, and the new nodes -- their weights come out sorted -- are merged with
            // the rest of Q into the next round's list (positions by binary search, old nodes first on equal weight).
            // 12-16 rounds for the 250-280 symbols of a BAM block instead of as many dependent steps on one lane
            // (126 k -> 36 k cycles per block; the same code lengths).
            u32* qw[2] = {L.wl, L.wi};
            u16* qi[2] = {L.qid[0], L.qid[1]};
            if (tid < n) L.qid[0][tid] = (u16)tid;                       // leaves 0 .. n-1 in weight order (rank step above)
            if (tid == 0) L.vars[V_K] = (u32)n;
            sync();
            u32 m = (u32)n, next_id = (u32)n;
            int cur = 0;
            while (m > 1) {
                const u32* w = qw[cur];
                const u32 t = w[0] + w[1];
                if ((u32)tid + 1 < m && w[tid] <= t && w[tid + 1] > t) L.vars[V_K] = (u32)tid + 1;     // at most one thread: Q is sorted
                sync();
                const u32 k = L.vars[V_K] & ~1u, a = m - k, b = k >> 1;
                if ((u32)tid < b) {                                      // new node tid of this round
                    const u32 nw = w[2 * tid] + w[2 * tid + 1];
                    const u32 id = next_id + (u32)tid;
                    L.parent[qi[cur][2 * tid]] = (u16)id; L.parent[qi[cur][2 * tid + 1]] = (u16)id;
                    u32 lo = 0, hi = a;                                  // old nodes (Q[k ..)) of weight <= nw go first
                    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (w[k + mid] <= nw) lo = mid + 1; else hi = mid; }
                    qw[cur ^ 1][(u32)tid + lo] = nw; qi[cur ^ 1][(u32)tid + lo] = (u16)id;
                } else if ((u32)tid >= k && (u32)tid < m) {              // old node that stays
                    const u32 ow = w[tid];
                    u32 lo = 0, hi = b;                                  // new nodes of weight < ow go first
                    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (w[2 * mid] + w[2 * mid + 1] < ow) lo = mid + 1; else hi = mid; }
                    qw[cur ^ 1][((u32)tid - k) + lo] = ow; qi[cur ^ 1][((u32)tid - k) + lo] = qi[cur][tid];
                }
                m = a + b; next_id += b; cur ^= 1;
                sync();
                if (tid == 0) L.vars[V_K] = m;                           // the next round's default: everything is <= t
                sync();
            }
        }
        sync();
        lap(10);
        if (tid < n) {                               // depth of leaf tid: steps to the root (node 2n - 2)
            int d = 0;
            for (int id = tid; id != 2 * n - 2; id = L.parent[id]) ++d;
            if (d > limit) L.vars[V_OVER] = 1;
            len[L.sorted[tid]] = (u8)d;
            atomicAdd(&L.bl_count[d < 16 ? d : 15], 1u);
        }
        sync();
        lap(11);
        if (!L.vars[V_OVER]) break;
        if (tid < N) { const u32 f = freq[tid]; if (f) freq[tid] = (f + 1) >> 1; }     // flatter, still >= 1
    }
    if (tid == 0) {
        u32 c = 0;
        L.next_code[0] = 0;
        L.bl_count[0] = 0;
        for (int bits = 1; bits <= 15; ++bits) { c = (c + L.bl_count[bits - 1]) << 1; L.next_code[bits] = c; }
    }
    sync();
    if (tid < N) {
        const u32 l = len[tid];
        u32 c = 0;
        if (l) {
            c = L.next_code[l];
            for (int j = 0; j < tid; ++j) c += (len[j] == l);
            c = __brev(c) >> (32 - l);
        }
        code[tid] = (u16)c;
    }
    __syncthreads();
    lap(12);
}

struct BitSink {          // LSB-first bit writer into 32-bit words shared with other writers (OR)
    u32* words; u32 word; u64 acc; u32 nbits;
    __device__ __forceinline__ void start(u32* w, u32 bitpos) { words = w; word = bitpos >> 5; acc = 0; nbits = bitpos & 31; }
    __device__ __forceinline__ void put(u32 v, u32 nb) {
        acc |= (u64)v << nbits;
        nbits += nb;
        if (nbits >= 32) { atomicOr(&words[word++], (u32)acc); acc >>= 32; nbits -= 32; }
    }
    __device__ __forceinline__ void finish() { if (nbits) atomicOr(&words[word], (u32)acc); }
};

// A token in 32 bits: bits 0..8 literal/length symbol (kTokNone: the position starts no token), 9..13 the length's
// extra value, 14..18 the distance symbol, 19..31 the distance's extra value.
constexpr u32 kTokNone = 0x1FFu;
__device__ __forceinline__ u32 length_extra_bits(u32 ls) { return (ls >= 265 && ls < 285) ? (ls - 261) >> 2 : 0u; }
__device__ __forceinline__ u32 dist_extra_bits(u32 ds) { return ds >= 4 ? (ds >> 1) - 1 : 0u; }
__device__ __forceinline__ u32 token_bits(const Lds& L, u32 tok) {
    const u32 ls = tok & 0x1FFu;
    if (ls == kTokNone) return 0;
    u32 nb = L.l_ll[ls];
    if (ls > 256) { const u32 ds = (tok >> 14) & 31u; nb += length_extra_bits(ls) + L.l_d[ds] + dist_extra_bits(ds); }
    return nb;
}

__global__ __launch_bounds__(kNT) void k_bgzf_deflate(DeflateArgs a) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    Lds& L = *reinterpret_cast<Lds*>(smem);
    const u32 tid = threadIdx.x;
    u32* const tokd = a.scratch + (size_t)blockIdx.x * kScratchPerWg;

    if (tid < 256) {
        u32 c = tid;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
        L.crc_tab[0][tid] = c;
    }
    __syncthreads();
    if (tid < 256) {
        u32 c = L.crc_tab[0][tid];
        for (int k = 1; k < 4; ++k) { c = (c >> 8) ^ L.crc_tab[0][c & 0xffu]; L.crc_tab[k][tid] = c; }
    }
    if (tid == 0) {
        u32 p = 0x40000000u;                   // x^1
        L.x2n[0] = p;
        for (int k = 1; k < 32; ++k) { p = multmodp(p, p); L.x2n[k] = p; }
    }
    __syncthreads();
    {
        // x^(512 j) for j < 1024 (what shifts a chunk's CRC over j later chunks of 64 bytes) and x^(8 r) for r <= 64 (over
        // the last, shorter chunk): products of the x^(2^k) above, made once per workgroup
        u32 tp = 0x80000000u;
        for (int k = 0; k < 10; ++k) if ((tid >> k) & 1u) tp = multmodp(L.x2n[9 + k], tp);
        L.tpow[tid] = tp;
        if (tid <= 64) {
            u32 xr = 0x80000000u;
            const u32 e = 8u * tid;
            for (int k = 0; k < 10; ++k) if ((e >> k) & 1u) xr = multmodp(L.x2n[k], xr);
            L.xr[tid] = xr;
        }
    }
    __syncthreads();

    for (u32 blk = blockIdx.x; blk < a.n_blocks; blk += gridDim.x) {
        const u64 o0 = a.off[blk];
        const u32 n = (u32)(a.off[blk + 1] - o0);
        const u32 mis = (u32)(o0 & 3u);
        u8* const in = reinterpret_cast<u8*>(L.buf) + mis;           // the LDS copy keeps the source's word alignment
        long long t_prev = a.prof ? clock64() : 0;
        auto lap = [&](int k) { if (a.prof && tid == 0) { const long long t = clock64(); atomicAdd(&a.prof[k], (unsigned long long)(t - t_prev)); t_prev = t; } };
        // ---- load (whole words; the bytes around the block are another block's or padding)
        {
            const u32* src = reinterpret_cast<const u32*>(a.in + (o0 - mis));
            const u32 nw = (n + mis + 3) >> 2;
            for (u32 w = tid; w < nw; w += kNT) L.buf[w] = src[w];
        }
        for (u32 i = tid; i < (1u << kHashBits); i += kNT) L.head[i] = 0;
        for (u32 i = tid; i < 288; i += kNT) L.f_ll[i] = 0;
        if (tid < 32) L.f_d[tid] = 0;
        if (tid < 20) L.f_cl[tid] = 0;
        __syncthreads();
        for (u32 i = tid; i < kPad; i += kNT) in[n + i] = 0;
        // byte histogram -> what a literal costs (a match is only worth taking if it beats the literals it replaces:
        // in quality strings of a few distinct values a literal costs 1-2 bits and a short match 15-25)
        // (every fourth byte is sample enough for an estimate)
        for (u32 i = tid * 4u + ((tid >> 8) & 3u); i < n; i += kNT * 4u) atomicAdd(&L.f_ll[in[i]], 1u);
        __syncthreads();
        if (tid < 256) {
            const u32 f = L.f_ll[tid];
            const float bits = f ? 16.0f * __log2f((float)((n + 3) >> 2) / (float)f) : 255.0f;
            L.bcost[tid] = (u8)min(255.0f, bits + 0.5f);
        }
        __syncthreads();
        if (tid < 256) L.f_ll[tid] = 0;
        lap(0);

        // ---- 1. match search.  Wavefront 0 walks the hash table for the NEXT window, 64 positions per step: the LDS
        //         executes a wavefront's instructions in order, so a step's lookups see every earlier step's entries and
        //         none of its own -- candidates are blind to the last < 64 bytes only, and the outcome is deterministic.
        if (tid == 0) { L.vars[V_NTOK] = 0; L.vars[V_CARRY] = 0; }
        const u32 plane = tid - kProd0;                  // the producer's lane (wavefront 15)
        auto produce = [&](u32 base, u16* cb) {
            // three passes so that the LDS sees the 15 lookup / insert pairs back to back (its in-order execution is what
            // orders them), instead of a wait for every lookup's result before the next pair is issued
            constexpr int kSteps = (int)kWin / 64;
            u32 h[kSteps], c[kSteps];
#pragma unroll
            for (int k = 0; k < kSteps; ++k) {
                const u32 p = base + (u32)k * 64u + plane;
                h[k] = p + 4 <= n ? (load32(in + p) * 2654435761u) >> (32 - kHashBits) : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int k = 0; k < kSteps; ++k) {
                const u32 p = base + (u32)k * 64u + plane;
                c[k] = 0;
                if (h[k] != 0xFFFFFFFFu) { c[k] = L.head[h[k]]; atomicMax(&L.head[h[k]], p + 1); }
            }
            // (two wavefronts walking the table, each owning the hash values of one parity, give the same candidates but were
            // measured slower: 545 k against 409 k cycles per block for the walk -- it is bound by LDS instructions, not lanes)
#pragma unroll
            for (int k = 0; k < kSteps; ++k) cb[k * 64 + (int)plane] = (u16)c[k];
        };
        __syncthreads();
        // kGrp windows per barrier: a wavefront that meets long matches in one window catches up in the next ones
        if (tid >= kProd0)
            for (u32 g = 0; g < kGrp; ++g) if (g * kWin < n) produce(g * kWin, L.cand[0] + g * kWin);
        __syncthreads();
        // Software pipeline over groups of kGrp windows (3840 positions = 60 chunks of 64), two workgroup barriers per group:
        //   phase A   wavefronts 0..14 decide the matches of group g (into gm[g & 1]) while wavefront 15 PARSES group g - 1
        //             (which positions start a token: a chain, see parse_group) and then looks up the hash candidates of
        //             group g + 1;
        //   phase B   every wavefront turns group g - 1's token starts into packed tokens (symbol counts in LDS, the tokens
        //             themselves densely to global memory -- the only per-position state that leaves the LDS; round 2 kept a
        //             256 KB match table per block in global scratch, written once and read twice: 16 x the algorithmic bytes).
        auto parse_group = [&](u32 gbase, const u32* gmv) {       // wavefront 14 only: lane l owns chunk l of the group
            const u32 l = tid - kParse0;
            const u32 glo = gbase + l * 64u;
            const u32 gcl = (l < kGrpChunks && glo < n) ? min(64u, n - glo) : 0u;
            u32 m2[32];                                      // the chunk's 64 token lengths (0 = literal), two per register
            if (l < kGrpChunks) {
                const uint4* v = reinterpret_cast<const uint4*>(gmv + l * 64u);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const uint4 x = v[q];
                    m2[2 * q] = (x.x >> 16) | (x.y & 0xffff0000u);
                    m2[2 * q + 1] = (x.z >> 16) | (x.w & 0xffff0000u);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 32; ++q) m2[q] = 0;
            }
#define MGX_LEN_AT(i) (((i) & 1) ? m2[(i) >> 1] >> 16 : m2[(i) >> 1] & 0xffffu)
            // The token at a position is fixed by the match phase; which positions START a token is the chain
            // carry -> carry + len -> ...  A lane first walks its chunk from the chunk's first position (in registers, 64
            // predicated steps), then learns where the previous chunk's last token really ends (lane 0: where the previous
            // group's did) and only re-walks from there UNTIL THE NEW CHAIN MEETS THE OLD ONE -- two chains through the same
            // jump table coincide from their first common position on, a few tokens in on BAM bytes --, reading the lengths
            // from the LDS table; repeated until no chunk's start moves.  One wavefront: lane shifts and ballots, no barrier.
            u64 mask = 0;                                    // token starts of the chunk, from cur_start
            u32 my_end = glo;                                // where the chunk's last token ends (absolute; may pass the chunk)
            {
                u32 next = 0;
#pragma unroll
                for (int i = 0; i < 64; ++i) if ((u32)i == next && (u32)i < gcl) { mask |= 1ull << i; const u32 ln = MGX_LEN_AT(i); next += ln ? ln : 1u; }
                if (gcl) my_end = glo + next;
            }
#undef MGX_LEN_AT
            const u32 carry = L.vars[V_CARRY];
            u32 cur_start = glo;
            for (;;) {
                const u32 up = (u32)__builtin_amdgcn_update_dpp(0, (int)my_end, 0x138, 0xf, 0xf, false);      // wave_shr:1: lane l reads lane l - 1
                const u32 prev = l == 0 ? carry : up;
                const u32 st = max(glo, prev);
                const bool changed = l < kGrpChunks && st != cur_start;
                if (changed) {
                    cur_start = st;
                    u32 nx = st - glo;                       // chunk-relative
                    u64 add = 0;
                    while (nx < gcl && !((mask >> nx) & 1ull)) {
                        add |= 1ull << nx;
                        const u32 ln = gmv[l * 64u + nx] >> 16;
                        nx += ln ? ln : 1u;
                    }
                    if (nx < gcl) mask = add | (mask & ~((1ull << nx) - 1ull));      // met the old chain: its tail (and its end) stand
                    else { mask = add; my_end = glo + nx; }                          // left the chunk first
                }
                if (!__ballot(changed)) break;
            }
            // tokens are stored densely, in position order: a chunk's tokens start at the count of all earlier ones
            const u32 cnt = l < kGrpChunks ? (u32)__builtin_popcountll(mask) : 0u;
            const u32 inc = wave_scan_inclusive(cnt, OpAdd());
            const u32 before = L.vars[V_NTOK];
            if (l < kGrpChunks) { L.mask[l] = mask; L.cend[l] = before + inc - cnt; }            // (group-local: the next group's parse comes after this group's tokens)
            const u32 all = wave_last(inc);
            const u32 last_end = (u32)__builtin_amdgcn_readlane((int)my_end, (int)kGrpChunks - 1);       // ends only grow along the chunks
            if (l == 0) { L.vars[V_NTOK] = before + all; L.vars[V_CARRY] = max(carry, last_end); }
        };
        auto tokens_of_group = [&](u32 gbase, const u32* gmv) {   // every thread; needs parse_group(gbase) behind a barrier
            for (u32 q = tid; q < kGrpPos; q += kNT) {
                const u32 p = gbase + q;
                if (p >= n) break;
                const u64 cmask = L.mask[q >> 6];             // (a group starts on a chunk boundary)
                if (!((cmask >> (q & 63u)) & 1ull)) continue;
                const u32 mm = gmv[q];
                const u32 len = mm >> 16;
                u32 t;
                if (len) {
                    u32 ls, leb, lev, ds, deb, dev;
                    length_code(len, &ls, &leb, &lev);
                    dist_code(mm & 0xffffu, &ds, &deb, &dev);
                    atomicAdd(&L.f_ll[ls], 1u);
                    atomicAdd(&L.f_d[ds], 1u);
                    t = ls | lev << 9 | ds << 14 | dev << 19;
                } else {
                    t = in[p];
                    atomicAdd(&L.f_ll[t], 1u);
                }
                tokd[L.cend[q >> 6] + (u32)__builtin_popcountll(cmask & ((1ull << (q & 63u)) - 1ull))] = t;
            }
        };
        u32 n_groups = 0;
        for (u32 grp = 0, base0 = 0; base0 < n; ++grp, base0 += kGrpPos) {
          n_groups = grp + 1;
          // ---- phase A
          const long long ta0 = a.prof ? clock64() : 0;
          if (tid >= kProd0) {
            for (u32 g = 0; g < kGrp; ++g) { const u32 b = base0 + (kGrp + g) * kWin; if (b < n) produce(b, L.cand[(grp + 1) & 1] + g * kWin); }
          } else if (tid >= kParse0) {
            if (grp) parse_group(base0 - kGrpPos, L.gm[(grp - 1) & 1]);
            if (a.prof && tid == kParse0) atomicAdd(&a.prof[13], (unsigned long long)(clock64() - ta0));
          } else for (u32 g = 0; g < kGrp; ++g) {
            const u32 base = base0 + g * kWin;
            if (base >= n) break;
            const u32 p = base + tid;
            u32 len = 0, dist = 0;
            // runs: byte p equals byte p - 1.  The wavefront's 64 flags in one ballot give every lane the length of the
            // distance-1 match at its position (its run of set flags) without a compare loop; only a run that reaches the
            // end of the wavefront's 64 positions is followed further
            const bool run_flag = p >= 1 && p < n && in[p - 1] == in[p];
            const u64 run_mask = __ballot(run_flag);
            // Hash candidates: inside a repeated stretch neighbouring positions point the same distance back, and then the
            // match at p + 1 is the match at p less its first byte.  Only the first lane of such a stretch (and lanes beyond
            // the end of their head's match) extend their candidate; the others take the head's length minus their offset.
            const u32 lane_ = tid & 63u;
            const u32 maxlen_ = p < n ? min(258u, n - p) : 0u;
            u32 cdist = 0;
            if (p < n) { const u32 cand = L.cand[grp & 1][g * kWin + tid]; if (cand && p - (cand - 1) <= 32768u) cdist = p - (cand - 1); }
            const u32 prev_dist = (u32)__builtin_amdgcn_update_dpp(0, (int)cdist, 0x138, 0xf, 0xf, false);      // wave_shr:1: lane i reads lane i - 1
            const bool head = cdist != 0 && (lane_ == 0 || cdist != prev_dist);
            u32 hlen = 0;
            if (head) hlen = match_len(in + p - cdist, in + p, maxlen_);
            const u64 head_mask = __ballot(head);
            const u64 below = head_mask & (lane_ == 63u ? ~0ull : ((2ull << lane_) - 1ull));
            const u32 head_lane = below ? 63u - (u32)__builtin_clzll(below) : 0u;
            const u32 head_len = (u32)__shfl((int)hlen, (int)head_lane, 64);
            u32 clen = hlen;
            if (cdist != 0 && !head) {
                const u32 off = lane_ - head_lane;
                clen = head_len > off ? head_len - off : match_len(in + p - cdist, in + p, maxlen_);     // past the head's match: on its own
            }
            if (p < n) {
                const u32 maxlen = maxlen_;
                if (clen >= 4) { len = clen; dist = cdist; }
                if (run_flag && maxlen >= 3) {
                    const u32 ln = tid & 63u;
                    const u64 rest = ~(run_mask >> ln);
                    u32 l = rest ? (u32)__builtin_ctzll(rest) : 64u;              // <= 64 - ln
                    if (ln + l == 64u && l < maxlen) l += match_len(in + p - 1 + l, in + p + l, maxlen - l);
                    l = min(l, maxlen);
                    if (l >= 3 && l >= len) { len = l; dist = 1; }
                }
                if (len && len <= 16 && a.cost_base) {
                    // the literals a short match replaces, priced from its first four bytes (three for a 3-byte match)
                    const u32 w4 = load32(in + p);
                    u32 lit = (u32)L.bcost[w4 & 0xffu] + L.bcost[(w4 >> 8) & 0xffu] + L.bcost[(w4 >> 16) & 0xffu];
                    lit = len == 3 ? lit : (lit + L.bcost[w4 >> 24]) * len >> 2;
                    const u32 x = dist - 1;
                    const u32 extra = x < 4 ? 0 : (31 - __builtin_clz(x)) - 1;
                    if (16u * ((dist == 1 ? a.cost_rle : a.cost_base) + extra) >= lit) len = 0;
                }
            }
            // one-step lazy evaluation: a longer match starting at the next byte (the neighbouring lane's) wins; the decision
            // is a function of the position alone, whatever the parse does around it
            const u32 len_next = (u32)__builtin_amdgcn_update_dpp(0, (int)len, 0x130, 0xf, 0xf, false);      // wave_shl:1: lane i reads lane i + 1
            if (a.lazy && (tid & 63u) != 63u && len_next > len) len = 0;
            L.gm[grp & 1][g * kWin + tid] = len ? (len << 16 | dist) : 0u;
          }
          if (a.prof && (tid == 0 || tid == kProd0)) atomicAdd(&a.prof[tid ? 15 : 14], (unsigned long long)(clock64() - ta0));
          __syncthreads();
          // ---- phase B
          const long long tb0 = a.prof ? clock64() : 0;
          if (grp) tokens_of_group(base0 - kGrpPos, L.gm[(grp - 1) & 1]);
          __syncthreads();
          if (a.prof && tid == 0) atomicAdd(&a.prof[7], (unsigned long long)(clock64() - tb0));
        }
        if (n_groups) {                                     // the last group's parse and tokens
            const u32 gb = (n_groups - 1) * kGrpPos;
            if (tid >= kParse0 && tid < kProd0) parse_group(gb, L.gm[(n_groups - 1) & 1]);
            __syncthreads();
            tokens_of_group(gb, L.gm[(n_groups - 1) & 1]);
            __syncthreads();
        }
        lap(1);

        const u32 lo = tid * 64u;
        const u32 cl = lo < n ? min(64u, n - lo) : 0u;
        // ---- 5. CRC-32 of the chunk, shifted to the end of the block
        {
            u32 c = tid == 0 ? 0xFFFFFFFFu : 0u;
            u32 p = lo;
            for (; p + 4 <= lo + cl; p += 4) {
                const u32 x = c ^ load32(in + p);
                c = L.crc_tab[3][x & 0xffu] ^ L.crc_tab[2][(x >> 8) & 0xffu] ^ L.crc_tab[1][(x >> 16) & 0xffu] ^ L.crc_tab[0][x >> 24];
            }
            for (; p < lo + cl; ++p) c = L.crc_tab[0][(c ^ in[p]) & 0xffu] ^ (c >> 8);
            // chunks 0 .. nc-2 are followed by r + 64 (nc - 2 - t) bytes (r = size of the last chunk): one multiplication
            // by the table's power here, the common factor x^(8 r) once after the reduction; the last chunk's state as it is
            const u32 nc = (n + 63u) >> 6;
            const u32 shifted = tid + 2u <= nc ? multmodp(L.tpow[nc - 2u - tid], c) : 0u;
            if (tid + 1u == (nc ? nc : 1u)) L.vars[V_CRCLAST] = c;
            const u32 part = wave_last(wave_scan_inclusive(shifted, OpXor()));
            if ((tid & 63u) == 0) L.wsum[tid >> 6] = part;
        }
        if (tid == 0) L.f_ll[256] = 1;             // end of block
        __syncthreads();
        if (tid == 0) {
            u32 c = 0;
            for (int i = 0; i < kNT / 64; ++i) c ^= L.wsum[i];
            const u32 nc = (n + 63u) >> 6;
            const u32 r = nc ? n - 64u * (nc - 1u) : 0u;
            L.vars[V_CRC] = multmodp(L.xr[r], c) ^ L.vars[V_CRCLAST] ^ 0xFFFFFFFFu;
        }
        lap(2);
        const u32 n_tok = L.vars[V_NTOK];
        lap(3);

        // ---- 3. codes
        huff_build<false>(L, L.f_ll, kNumLL, 15, L.l_ll, L.c_ll, a.prof);
        huff_build<true>(L, L.f_d, kNumD, 15, L.l_d, L.c_d);
        lap(4);
        // the header (RFC 1951 3.2.7), in parallel: the nlit + ndist code lengths are cut into runs (ballots), every run
        // knows how many code-length symbols it becomes (16: repeat 3-6, 17 / 18: zeros 3-10 / 11-138), a scan places them
        if (tid == 0) { L.vars[V_NLIT] = 257; L.vars[V_NDIST] = 1; }
        for (u32 i = tid; i < 192; i += kNT) L.hdr[i] = 0;
        __syncthreads();
        if (tid < (u32)kNumLL && L.l_ll[tid]) atomicMax(&L.vars[V_NLIT], tid + 1);
        if (tid < (u32)kNumD && L.l_d[tid]) atomicMax(&L.vars[V_NDIST], tid + 1);
        __syncthreads();
        {
            const u32 nlit = L.vars[V_NLIT], total = nlit + L.vars[V_NDIST];
            auto length_at = [&](u32 i) -> u32 { return i < nlit ? L.l_ll[i] : L.l_d[i - nlit]; };
            const u32 v = tid < total ? length_at(tid) : 0xFFu;
            const bool start = tid < total && (tid == 0 || length_at(tid - 1) != v);
            const u64 bal = __ballot(start);
            if ((tid & 63u) == 0 && tid < 384) L.smask[tid >> 6] = bal;
            __syncthreads();
            u32 cnt = 0, run = 0;
            if (start) {
                u32 next = total;
                const u32 wv = tid >> 6, ln = tid & 63u;
                const u64 rest = ln == 63 ? 0ull : (L.smask[wv] >> (ln + 1));
                if (rest) next = tid + 1 + (u32)__builtin_ctzll(rest);
                else for (u32 w = wv + 1; w < 6; ++w) { const u64 mk = L.smask[w]; if (mk) { next = w * 64 + (u32)__builtin_ctzll(mk); break; } }
                if (next > total) next = total;
                run = next - tid;
                if (v == 0) { const u32 rem = run % 138; cnt = run / 138 + (rem >= 3 ? 1u : rem); }
                else { const u32 rem = (run - 1) % 6; cnt = 1 + (run - 1) / 6 + (rem >= 3 ? 1u : rem); }
            }
            u32 ns;
            u32 at = block_scan(L, cnt, &ns);
            if (start) {
                auto emit = [&](u32 sym, u32 extra) { L.clsym[at++] = (u16)(sym | extra << 8); atomicAdd(&L.f_cl[sym], 1u); };
                if (v == 0) {
                    u32 r = run;
                    for (; r >= 138; r -= 138) emit(18, 127);
                    if (r >= 11) emit(18, r - 11); else if (r >= 3) emit(17, r - 3); else for (; r; --r) emit(0, 0);
                } else {
                    emit(v, 0);
                    u32 r = run - 1;
                    for (; r >= 6; r -= 6) emit(16, 3);
                    if (r >= 3) emit(16, r - 3); else for (; r; --r) emit(v, 0);
                }
            }
            if (tid == 0) L.vars[V_NCLSYM] = ns;
        }
        huff_build<true>(L, L.f_cl, kNumCL, 7, L.l_cl, L.c_cl);
        if (tid == 0) {
            const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            u32 word = 0, nbits = 0; u64 acc = 0;
            auto put = [&](u32 v, u32 nb) {
                acc |= (u64)v << nbits; nbits += nb;
                if (nbits >= 32) { atomicOr(&L.hdr[word++], (u32)acc); acc >>= 32; nbits -= 32; }
            };
            int ncl = 19;
            while (ncl > 4 && L.l_cl[order[ncl - 1]] == 0) --ncl;
            put(1u | 2u << 1, 3);                                   // BFINAL = 1, BTYPE = 10 (dynamic)
            put(L.vars[V_NLIT] - 257, 5); put(L.vars[V_NDIST] - 1, 5); put((u32)ncl - 4, 4);
            for (int i = 0; i < ncl; ++i) put(L.l_cl[order[i]], 3);
            L.vars[V_HDRBITS] = word * 32 + nbits;                  // so far; the code-length symbols follow
            if (nbits) atomicOr(&L.hdr[word], (u32)acc);
        }
        __syncthreads();
        {
            const u32 ns = L.vars[V_NCLSYM], fixed_bits = L.vars[V_HDRBITS];
            u32 sym = 0, extra = 0, nb = 0;
            if (tid < ns) {
                sym = L.clsym[tid] & 0xffu; extra = L.clsym[tid] >> 8;
                nb = L.l_cl[sym] + (sym == 16 ? 2u : sym == 17 ? 3u : sym == 18 ? 7u : 0u);
            }
            u32 sym_bits;
            const u32 off = block_scan(L, nb, &sym_bits);
            if (tid < ns) {
                BitSink bs;
                bs.start(L.hdr, fixed_bits + off);
                bs.put(L.c_cl[sym], L.l_cl[sym]);
                if (sym == 16) bs.put(extra, 2); else if (sym == 17) bs.put(extra, 3); else if (sym == 18) bs.put(extra, 7);
                bs.finish();
            }
            __syncthreads();
            if (tid == 0) L.vars[V_HDRBITS] = fixed_bits + sym_bits;
        }
        __syncthreads();
        lap(5);

        // ---- 4. size (from the symbol counts), then the bits, position-parallel
        u32 sym_bits = 0;
        if (tid < (u32)kNumLL) sym_bits = L.f_ll[tid] * (L.l_ll[tid] + (tid > 256 ? length_extra_bits(tid) : 0u));
        else if (tid >= 512 && tid < 512u + kNumD) sym_bits = L.f_d[tid - 512] * (L.l_d[tid - 512] + dist_extra_bits(tid - 512));
        u32 token_total;
        (void)block_scan(L, sym_bits, &token_total);          // includes the end-of-block symbol
        const u32 hdr_bits = L.vars[V_HDRBITS];
        const u32 total_bits = hdr_bits + token_total;
        u32 payload = (total_bits + 7) >> 3;
        const bool stored = payload >= n + 5 || n == 0;
        u8* const slot = a.slots + (size_t)blk * kSlot + kSlotSkew;
        u8* const pay = slot + 18;
        __syncthreads();                                    // every reader of the input bytes is done
        if (!stored) {
            const u32 nw = (payload + 3) >> 2;
            for (u32 w = tid; w < nw; w += kNT) L.buf[w] = 0;
            __syncthreads();
            for (u32 w = tid; w < ((hdr_bits + 31) >> 5); w += kNT) atomicOr(&L.buf[w], L.hdr[w]);
            // The tokens are read ONCE, a window of 1024 at a time: a window's bit offsets are the bits of the earlier windows
            // (carried in a register) plus a workgroup scan of its own tokens' sizes; the next window's tokens are already in
            // flight during the scan.  (Round 2 read the token list twice: once for the sizes, once for the bits.)
            const u32 n_tw = (n_tok + kNT - 1) / kNT;
            u32 all_bits = 0;
            u32 tnext = tid < n_tok ? tokd[tid] : kTokNone;
            for (u32 w = 0; w < n_tw; ++w) {
                const u32 t = tnext;
                const u32 inext = (w + 1) * kNT + tid;
                tnext = inext < n_tok ? tokd[inext] : kTokNone;
                const u32 nb = token_bits(L, t);
                u32 win_bits;
                const u32 off = block_scan(L, nb, &win_bits);
                if (nb) {
                    BitSink sk;
                    sk.start(L.buf, hdr_bits + all_bits + off);
                    const u32 ls = t & 0x1FFu;
                    sk.put(L.c_ll[ls], L.l_ll[ls]);
                    if (ls > 256) {
                        const u32 ds = (t >> 14) & 31u, leb = length_extra_bits(ls), deb = dist_extra_bits(ds);
                        if (leb) sk.put((t >> 9) & 31u, leb);
                        sk.put(L.c_d[ds], L.l_d[ds]);
                        if (deb) sk.put(t >> 19, deb);
                    }
                    sk.finish();
                }
                all_bits += win_bits;
            }
            const u32 carry = hdr_bits + all_bits;
            if (tid == 0) { BitSink s; s.start(L.buf, carry); s.put(L.c_ll[256], L.l_ll[256]); s.finish(); }
            __syncthreads();
            u32* payw = reinterpret_cast<u32*>(pay);
            for (u32 w = tid; w < (payload >> 2); w += kNT) payw[w] = L.buf[w];
            if (tid == 0) {
                const u8* ob = reinterpret_cast<const u8*>(L.buf);
                for (u32 i = payload & ~3u; i < payload; ++i) pay[i] = ob[i];
            }
        } else {
            payload = n + 5;
            if (tid == 0) {
                pay[0] = 1; pay[1] = (u8)(n & 0xff); pay[2] = (u8)(n >> 8); pay[3] = (u8)(~n & 0xff); pay[4] = (u8)((~n >> 8) & 0xff);
                atomicAdd(a.n_stored, 1u);
            }
            for (u32 i = tid; i < n; i += kNT) pay[5 + i] = in[i];
        }
        if (tid == 0) {
            const u32 total = 18 + payload + 8;
            const u8 head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
            for (int i = 0; i < 16; ++i) slot[i] = head[i];
            slot[16] = (u8)((total - 1) & 0xff); slot[17] = (u8)((total - 1) >> 8);
            const u32 crc = L.vars[V_CRC];
            u8* f = pay + payload;
            for (int i = 0; i < 4; ++i) { f[i] = (u8)(crc >> (8 * i)); f[4 + i] = (u8)(n >> (8 * i)); }
            a.sizes[blk] = total;
        }
        __syncthreads();                                    // LDS is reused by the next block
        lap(6);
    }
}

// exclusive prefix sum of the block sizes (one workgroup; a batch has a few thousand blocks)
__global__ __launch_bounds__(256) void k_bgzf_offsets(const u32* sizes, u32 n, u64* out_off) {
    __shared__ u64 part[256];
    const u32 tid = threadIdx.x;
    const u32 per = (n + 255) / 256;
    const u32 lo = min(n, tid * per), hi = min(n, lo + per);
    u64 s = 0;
    for (u32 i = lo; i < hi; ++i) s += sizes[i];
    part[tid] = s;
    __syncthreads();
    if (tid == 0) { u64 run = 0; for (int i = 0; i < 256; ++i) { const u64 v = part[i]; part[i] = run; run += v; } out_off[n] = run; }
    __syncthreads();
    u64 run = part[tid];
    for (u32 i = lo; i < hi; ++i) { out_off[i] = run; run += sizes[i]; }
}

// slot -> its place in the packed output (pinned host memory: the stores cross PCIe).  A bounded grid of small workgroups
// walks the blocks: a workgroup per block would fill every CU's wavefront slots with stores waiting on the link and keep the
// next batch's 1024-thread deflate workgroups from being placed.
__global__ __launch_bounds__(256) void k_bgzf_pack(const u8* slots, const u32* sizes, const u64* out_off, u8* out, u32 n_blocks) {
    for (u32 blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const u8* src = slots + (size_t)blk * kSlot + kSlotSkew;
        u8* dst = out + out_off[blk];
        const u32 n = sizes[blk];
        // destination-aligned words, source read with whatever alignment it has
        const u32 lead = min(n, (u32)((4u - ((uintptr_t)dst & 3u)) & 3u));
        if (threadIdx.x < lead) dst[threadIdx.x] = src[threadIdx.x];
        const u32 nw = (n - lead) >> 2;
        u32* dw = reinterpret_cast<u32*>(dst + lead);
        const u8* sb = src + lead;
        for (u32 w = threadIdx.x; w < nw; w += 256) dw[w] = load32(sb + 4 * (size_t)w);
        const u32 done = lead + 4 * nw;
        if (threadIdx.x < n - done) dst[done + threadIdx.x] = src[done + threadIdx.x];
    }
}

// ---- the record store: records resident in HBM, emitted in sorted order ------------------------------------------
constexpr u32 kScanTile = 4096;
// slen[q] = 4 + len[order[q]] summed per tile of 4096 records
__global__ __launch_bounds__(256) void k_store_tile_sums(const u32* order, const u32* len, u64 n, u64* tile_sum) {
    __shared__ u64 part[256];
    const u64 base = (u64)blockIdx.x * kScanTile;
    u64 s = 0;
    for (u32 i = threadIdx.x; i < kScanTile; i += 256) { const u64 q = base + i; if (q < n) s += 4u + len[order[q]]; }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) { if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = part[0];
}
__global__ __launch_bounds__(1024) void k_store_scan_tiles(u64* tile_sum, u32 n_tiles, u64* total) {      // one workgroup, in place, exclusive
    __shared__ u64 part[1024];
    const u32 per = (n_tiles + 1023) / 1024;
    const u32 lo = min(n_tiles, threadIdx.x * per), hi = min(n_tiles, lo + per);
    u64 s = 0;
    for (u32 i = lo; i < hi; ++i) s += tile_sum[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { u64 run = 0; for (int i = 0; i < 1024; ++i) { const u64 v = part[i]; part[i] = run; run += v; } *total = run; }
    __syncthreads();
    u64 run = part[threadIdx.x];
    for (u32 i = lo; i < hi; ++i) { const u64 v = tile_sum[i]; tile_sum[i] = run; run += v; }
}
// uoff[q] = offset of record q (its block_size field) in the uncompressed stream
__global__ __launch_bounds__(256) void k_store_offsets(const u32* order, const u32* len, u64 n, const u64* tile_base, u64* uoff) {
    __shared__ u64 part[256];
    const u64 base = (u64)blockIdx.x * kScanTile + (u64)threadIdx.x * 16u;
    u32 l[16];
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const u64 q = base + i; l[i] = q < n ? 4u + len[order[q]] : 0u; s += l[i]; }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { u64 run = tile_base[blockIdx.x]; for (int i = 0; i < 256; ++i) { const u64 v = part[i]; part[i] = run; run += v; } }
    __syncthreads();
    u64 run = part[threadIdx.x];
#pragma unroll
    for (int i = 0; i < 16; ++i) { const u64 q = base + i; if (q < n) uoff[q] = run; run += l[i]; }
}
// One wavefront per record q in [q0, q1): block_size + bytes to their place in the window [win0, win0 + win_bytes) of the
// stream; a duplicate gets FLAG |= 0x400 (byte 15 of the record, bit 2) on the way.
__global__ __launch_bounds__(256) void k_store_gather(const u64* addr, const u32* len, const u32* order, const u8* dup, const u64* uoff,
                                                      u64 q0, u64 q1, u64 win0, u64 win_bytes, u8* dst) {
    const u64 q = q0 + (u64)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (q >= q1) return;
    const u32 lane = threadIdx.x & 63u;
    const u32 r = order[q];
    const u32 n = len[r];
    const u8* src = reinterpret_cast<const u8*>(addr[r]);
    const u64 at = uoff[q];
    const bool is_dup = dup[r] != 0;
    for (u32 i = lane; i < n + 4u; i += 64u) {
        const u64 o = at + i;
        if (o < win0 || o >= win0 + win_bytes) continue;
        u8 v;
        if (i < 4) v = (u8)(n >> (8 * i));
        else { v = src[i - 4]; if (is_dup && i == 4u + 15u) v |= 0x04; }
        dst[o - win0] = v;
    }
}

}  // namespace

struct mgx_bgzf {
    int device = 0;
    int n_cu = 0;
    hipStream_t stream = nullptr;              // all batches of a context run in order on one stream
    hipStream_t copy = nullptr;                // packed blocks travel back on a stream of their own, behind their batch's kernels only
    hipStream_t up = nullptr;                  // ... and a batch's input travels up on a third one, under the kernels of the batch before
    std::mutex prep_mu; bool prepared = false; // the compressor's scratch and kernel attributes: set up at first use (or mgx_bgzf_prepare)
    u32* d_scratch = nullptr; u32 grid = 0;
    u32* d_n_stored = nullptr;
    unsigned long long* d_prof = nullptr;
    u32 lazy = 1, cost_base = 10, cost_rle = 6;
    u64 n_blocks = 0, bytes_in = 0, bytes_out = 0;
    float ms_kernels = 0, ms_pack = 0;
};

struct mgx_bgzf_batch {
    u64 in_cap = 0; u32 max_blocks = 0;
    u8* h_in = nullptr; u64* h_off = nullptr;          // pinned, filled by the caller
    u8* h_out = nullptr; u64* h_out_off = nullptr;     // pinned, results
    u8* d_in = nullptr; u64* d_off = nullptr; u8* d_slots = nullptr; u32* d_sizes = nullptr; u64* d_out_off = nullptr;
    u64 out_cap = 0;
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_p0 = nullptr, ev_out = nullptr, ev_in = nullptr;
    u8* h_out_dev = nullptr;                          // the pinned output buffer as the device addresses it
    u32 n_blocks = 0; u64 n_in = 0;
    bool submitted = false;
};

extern "C" {

uint64_t mgx_bgzf_bound(uint64_t n_bytes, uint64_t n_blocks) { return n_bytes + 31 * n_blocks; }

int mgx_bgzf_create(int device, unsigned flags, mgx_bgzf_t** out) {
    if (!out) { set_error("out is NULL"); return -EINVAL; }
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { set_error("no HIP device: the BGZF compressor has no CPU path"); return -ENODEV; }
    if (device < 0) device = 0;
    if (device >= n_dev) { set_error("device %d of %d", device, n_dev); return -EINVAL; }
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<mgx_bgzf> c(new (std::nothrow) mgx_bgzf);
    if (!c) { set_error("out of memory"); return -ENOMEM; }
    c->device = device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount;
    c->lazy = (flags & 1u) ? 0 : 1;
    if (const char* e = getenv("MGX_BGZF_LAZY")) c->lazy = atoi(e) != 0;
    if (const char* e = getenv("MGX_BGZF_COST_BASE")) c->cost_base = (u32)atoi(e);
    if (const char* e = getenv("MGX_BGZF_COST_RLE")) c->cost_rle = (u32)atoi(e);
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->up, hipStreamNonBlocking));
    // one workgroup per CU (157 KB of LDS, 1024 threads x 128 registers: a CU that runs one has no register left for anything
    // else, so the pack kernel of the batch before cannot run BESIDE a deflate kernel, only between two).  Leaving 8 or 16 CUs
    // free for it was measured: the deflate kernel then takes 9 rounds instead of 8 over a 2048-block batch (3.39 against
    // 3.04 ms) and pinned -> pinned falls from 28.0 to 24.1 GB/s.
    c->grid = (u32)c->n_cu;
    if (const char* e = getenv("MGX_BGZF_GRID")) { const int v = atoi(e); if (v > 0) c->grid = (u32)v; }
    // the compressor's own device state (scratch tables, counters, the kernel's LDS attribute -- which loads the code
    // object) is set up by the first batch or by mgx_bgzf_prepare: a record store needs none of it, and a tool that
    // creates the context while it parses its first input should not wait for it
    *out = c.release();
    return 0;
}

int mgx_bgzf_prepare(mgx_bgzf_t* c) {
    if (!c) { set_error("NULL argument"); return -EINVAL; }
    std::lock_guard<std::mutex> g(c->prep_mu);
    if (c->prepared) return 0;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc((void**)&c->d_scratch, (size_t)c->grid * kScratchPerWg * sizeof(u32)));
    HIP_TRY(hipMalloc((void**)&c->d_n_stored, sizeof(u32)));
    HIP_TRY(hipMemsetAsync(c->d_n_stored, 0, sizeof(u32), c->stream));
    if (const char* e = getenv("MGX_BGZF_PROF")) if (atoi(e)) {
        HIP_TRY(hipMalloc((void**)&c->d_prof, 16 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(c->d_prof, 0, 16 * sizeof(unsigned long long), c->stream));
    }
    HIP_TRY(hipFuncSetAttribute((const void*)k_bgzf_deflate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds)));
    c->prepared = true;
    return 0;
}

int mgx_bgzf_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes) {
    if (!free_bytes || !total_bytes) { set_error("NULL argument"); return -EINVAL; }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { set_error("no HIP device"); return -ENODEV; }
    if (device < 0) device = 0;
    if (device >= n_dev) { set_error("device %d of %d", device, n_dev); return -EINVAL; }
    HIP_TRY(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    *free_bytes = f; *total_bytes = t;
    return 0;
}

void mgx_bgzf_destroy(mgx_bgzf_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->copy) { (void)hipStreamSynchronize(c->copy); (void)hipStreamDestroy(c->copy); }
    if (c->up) { (void)hipStreamSynchronize(c->up); (void)hipStreamDestroy(c->up); }
    (void)hipFree(c->d_scratch);
    (void)hipFree(c->d_n_stored);
    if (c->d_prof) {
        unsigned long long p[16];
        if (hipMemcpy(p, c->d_prof, sizeof p, hipMemcpyDeviceToHost) == hipSuccess && c->n_blocks) {
            fprintf(stderr, "mgx_bgzf cycles per block: load %llu, match %llu, parse+crc %llu, tokens %llu, huffman %llu, header %llu, emit %llu\n", p[0] / c->n_blocks,
                    p[1] / c->n_blocks, p[2] / c->n_blocks, p[3] / c->n_blocks, p[4] / c->n_blocks, p[5] / c->n_blocks, p[6] / c->n_blocks);
            fprintf(stderr, "   parse of the groups on wavefront 14 (inside phase A): %llu\n", p[13] / c->n_blocks);
            fprintf(stderr, "   inside match: phase A as wavefront 0 sees it %llu, as wavefront 15 (hash candidates) %llu; phase B (tokens) %llu\n", p[14] / c->n_blocks, p[15] / c->n_blocks, p[7] / c->n_blocks);
            fprintf(stderr, "   literal/length code: setup %llu, rank %llu, merge rounds %llu, depths %llu, codes %llu\n", p[8] / c->n_blocks, p[9] / c->n_blocks, p[10] / c->n_blocks, p[11] / c->n_blocks, p[12] / c->n_blocks);
        }
        (void)hipFree(c->d_prof);
    }
    delete c;
}

void mgx_bgzf_batch_destroy(mgx_bgzf_t* c, mgx_bgzf_batch_t* b) {
    if (!b) return;
    if (c) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->up); (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->copy); }
    (void)hipHostFree(b->h_in); (void)hipHostFree(b->h_off); (void)hipHostFree(b->h_out); (void)hipHostFree(b->h_out_off);
    (void)hipFree(b->d_in); (void)hipFree(b->d_off); (void)hipFree(b->d_slots); (void)hipFree(b->d_sizes); (void)hipFree(b->d_out_off);
    if (b->ev_k0) (void)hipEventDestroy(b->ev_k0);
    if (b->ev_k1) (void)hipEventDestroy(b->ev_k1);
    if (b->ev_p0) (void)hipEventDestroy(b->ev_p0);
    if (b->ev_out) (void)hipEventDestroy(b->ev_out);
    if (b->ev_in) (void)hipEventDestroy(b->ev_in);
    delete b;
}

static int batch_create(mgx_bgzf_t* c, uint64_t in_capacity, uint32_t max_blocks, bool pinned_input, mgx_bgzf_batch_t** out);
int mgx_bgzf_batch_create(mgx_bgzf_t* c, uint64_t in_capacity, uint32_t max_blocks, mgx_bgzf_batch_t** out) {
    return batch_create(c, in_capacity, max_blocks, true, out);
}
static int batch_create(mgx_bgzf_t* c, uint64_t in_capacity, uint32_t max_blocks, bool pinned_input, mgx_bgzf_batch_t** out) {
    if (!c || !out) { set_error("NULL argument"); return -EINVAL; }
    *out = nullptr;
    if (max_blocks == 0) { set_error("max_blocks is 0"); return -EINVAL; }
    in_capacity = std::min<u64>(in_capacity, (u64)max_blocks * kMaxIn);
    if (const int prc = mgx_bgzf_prepare(c)) return prc;
    HIP_TRY(hipSetDevice(c->device));
    mgx_bgzf_batch* b = new (std::nothrow) mgx_bgzf_batch;
    if (!b) { set_error("out of memory"); return -ENOMEM; }
    b->in_cap = in_capacity; b->max_blocks = max_blocks;
    b->out_cap = mgx_bgzf_bound(in_capacity, max_blocks);
    auto fail = [&](const char* what) { set_error("%s failed for a batch of %llu bytes / %u blocks", what, (unsigned long long)in_capacity, max_blocks); mgx_bgzf_batch_destroy(c, b); return -ENOMEM; };
    if (pinned_input && hipHostMalloc((void**)&b->h_in, in_capacity + 8, hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc");
    if (hipHostMalloc((void**)&b->h_off, ((size_t)max_blocks + 1) * sizeof(u64), hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc");
    if (hipHostMalloc((void**)&b->h_out, b->out_cap, hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc");
    if (hipHostMalloc((void**)&b->h_out_off, ((size_t)max_blocks + 1) * sizeof(u64), hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc");
    if (hipMalloc((void**)&b->d_in, in_capacity + 8) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&b->d_off, ((size_t)max_blocks + 1) * sizeof(u64)) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&b->d_slots, (size_t)max_blocks * kSlot) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&b->d_sizes, (size_t)max_blocks * sizeof(u32)) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&b->d_out_off, ((size_t)max_blocks + 1) * sizeof(u64)) != hipSuccess) return fail("hipMalloc");
    if (hipHostGetDevicePointer((void**)&b->h_out_dev, b->h_out, 0) != hipSuccess) return fail("hipHostGetDevicePointer");
    if (hipEventCreate(&b->ev_k0) != hipSuccess || hipEventCreate(&b->ev_k1) != hipSuccess ||
        hipEventCreate(&b->ev_p0) != hipSuccess ||
        hipEventCreate(&b->ev_out) != hipSuccess ||
        hipEventCreateWithFlags(&b->ev_in, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate");
    b->h_off[0] = 0;
    *out = b;
    return 0;
}

uint8_t* mgx_bgzf_batch_input(mgx_bgzf_batch_t* b) { return b ? b->h_in : nullptr; }
uint64_t* mgx_bgzf_batch_offsets(mgx_bgzf_batch_t* b) { return b ? b->h_off : nullptr; }

static int batch_submit(mgx_bgzf_t* c, mgx_bgzf_batch_t* b, uint32_t n_blocks, bool input_resident);
int mgx_bgzf_batch_submit(mgx_bgzf_t* c, mgx_bgzf_batch_t* b, uint32_t n_blocks) { return batch_submit(c, b, n_blocks, false); }
// input_resident: the batch's device input buffer was filled by a kernel on the context's stream (the record store)
static int batch_submit(mgx_bgzf_t* c, mgx_bgzf_batch_t* b, uint32_t n_blocks, bool input_resident) {
    if (!c || !b) { set_error("NULL argument"); return -EINVAL; }
    if (n_blocks > b->max_blocks) { set_error("%u blocks in a batch made for %u", n_blocks, b->max_blocks); return -EINVAL; }
    if (b->h_off[0] != 0) { set_error("offsets[0] must be 0"); return -EINVAL; }
    for (u32 i = 0; i < n_blocks; ++i) {
        if (b->h_off[i + 1] < b->h_off[i] || b->h_off[i + 1] - b->h_off[i] > kMaxIn) {
            set_error("block %u: [%llu, %llu) is not a piece of at most %u bytes", i, (unsigned long long)b->h_off[i], (unsigned long long)b->h_off[i + 1], kMaxIn);
            return -EINVAL;
        }
    }
    const u64 n_in = b->h_off[n_blocks];
    if (n_in > b->in_cap) { set_error("%llu input bytes in a batch made for %llu", (unsigned long long)n_in, (unsigned long long)b->in_cap); return -EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    b->n_blocks = n_blocks; b->n_in = n_in; b->submitted = true;
    if (n_blocks == 0) { b->h_out_off[0] = 0; return 0; }
    hipStream_t s = c->stream;
    if (!input_resident) {
        // the input goes up on its own stream: on the kernel stream it would wait for the deflate kernels of the batches
        // submitted before (d_in was last read by this batch's previous kernels, which its last wait() has seen finish)
        HIP_TRY(hipMemcpyAsync(b->d_in, b->h_in, n_in, hipMemcpyHostToDevice, c->up));
        HIP_TRY(hipEventRecord(b->ev_in, c->up));
        HIP_TRY(hipStreamWaitEvent(s, b->ev_in, 0));
    }
    HIP_TRY(hipMemcpyAsync(b->d_off, b->h_off, ((size_t)n_blocks + 1) * sizeof(u64), hipMemcpyHostToDevice, s));
    DeflateArgs a{};
    a.in = b->d_in; a.off = b->d_off; a.n_blocks = n_blocks; a.slots = b->d_slots; a.sizes = b->d_sizes;
    a.scratch = c->d_scratch; a.n_stored = c->d_n_stored; a.lazy = c->lazy; a.cost_base = c->cost_base; a.cost_rle = c->cost_rle; a.prof = c->d_prof;
    HIP_TRY(hipEventRecord(b->ev_k0, s));
    hipLaunchKernelGGL(k_bgzf_deflate, dim3(std::min(c->grid, n_blocks)), dim3(kNT), sizeof(Lds), s, a);
    HIP_TRY(hipEventRecord(b->ev_k1, s));
    // The finished blocks are packed STRAIGHT INTO THE PINNED OUTPUT BUFFER by the pack kernel, on the copy stream behind this
    // batch's deflate kernel only: no packed copy in HBM and no device-to-host copy behind it (round 2's was carried out by
    // a blit kernel that took 4.3 ms beside the next batch's deflate kernel and slowed that one from 3.0 to 4.1 ms:
    // 20 GB/s pinned to pinned; tools trace in profiles/r03_bgzf_pipeline_trace.txt), and the next batch's deflate kernel
    // does not queue behind the packing.
    HIP_TRY(hipStreamWaitEvent(c->copy, b->ev_k1, 0));
    HIP_TRY(hipEventRecord(b->ev_p0, c->copy));
    hipLaunchKernelGGL(k_bgzf_offsets, dim3(1), dim3(256), 0, c->copy, b->d_sizes, n_blocks, b->d_out_off);
    hipLaunchKernelGGL(k_bgzf_pack, dim3(std::min<u32>(n_blocks, (u32)c->n_cu)), dim3(256), 0, c->copy, b->d_slots, b->d_sizes, b->d_out_off, b->h_out_dev, n_blocks);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(b->h_out_off, b->d_out_off, ((size_t)n_blocks + 1) * sizeof(u64), hipMemcpyDeviceToHost, c->copy));
    HIP_TRY(hipEventRecord(b->ev_out, c->copy));
    return 0;
}

int mgx_bgzf_batch_wait(mgx_bgzf_t* c, mgx_bgzf_batch_t* b, const uint8_t** out, const uint64_t** out_offsets) {
    if (!c || !b || !out || !out_offsets) { set_error("NULL argument"); return -EINVAL; }
    if (!b->submitted) { set_error("batch was not submitted"); return -EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    *out = b->h_out; *out_offsets = b->h_out_off;
    b->submitted = false;
    if (b->n_blocks == 0) return 0;
    HIP_TRY(hipEventSynchronize(b->ev_out));
    const u64 total = b->h_out_off[b->n_blocks];
    if (total > b->out_cap) { set_error("internal: %llu output bytes exceed the bound %llu", (unsigned long long)total, (unsigned long long)b->out_cap); return -EIO; }
    float ms = 0, ms_pack = 0;
    if (hipEventElapsedTime(&ms, b->ev_k0, b->ev_k1) == hipSuccess && hipEventElapsedTime(&ms_pack, b->ev_p0, b->ev_out) == hipSuccess)
    { c->ms_kernels = ms; c->ms_pack = ms_pack; }     // the deflate kernel | offsets + pack (its stores cross PCIe) + offsets read-back on the copy stream
    c->n_blocks += b->n_blocks; c->bytes_in += b->n_in; c->bytes_out += total;
    return 0;
}

int mgx_bgzf_compress(mgx_bgzf_t* c, const uint8_t* in, const uint64_t* offsets, uint64_t n_blocks, uint8_t* out,
                      uint64_t out_capacity, uint64_t* out_offsets) {
    if (!c || !offsets || !out_offsets || (n_blocks && (!in || !out))) { set_error("NULL argument"); return -EINVAL; }
    const u64 base = offsets[0];
    const u64 n_bytes = offsets[n_blocks] - base;
    if (out_capacity < mgx_bgzf_bound(n_bytes, n_blocks)) { set_error("output capacity %llu is below mgx_bgzf_bound = %llu", (unsigned long long)out_capacity, (unsigned long long)mgx_bgzf_bound(n_bytes, n_blocks)); return -EINVAL; }
    out_offsets[0] = 0;
    if (n_blocks == 0) return 0;
    constexpr u32 kPer = 1024;                  // blocks per internal batch (64 MB)
    const u32 per = (u32)std::min<u64>(kPer, n_blocks);
    mgx_bgzf_batch_t* bt[2] = {nullptr, nullptr};
    int rc = 0;
    for (int i = 0; i < 2 && !rc; ++i) rc = mgx_bgzf_batch_create(c, (u64)per * kMaxIn, per, &bt[i]);
    u64 done_out = 0;
    struct Flight { u64 first, count; };
    Flight fl[2] = {{0, 0}, {0, 0}};
    auto drain = [&](int k) -> int {
        if (!fl[k].count) return 0;
        const uint8_t* o; const uint64_t* oo;
        const int r = mgx_bgzf_batch_wait(c, bt[k], &o, &oo);
        if (r) return r;
        memcpy(out + done_out, o, oo[fl[k].count]);
        for (u64 i = 0; i < fl[k].count; ++i) out_offsets[fl[k].first + i + 1] = done_out + oo[i + 1];
        done_out += oo[fl[k].count];
        fl[k].count = 0;
        return 0;
    };
    int k = 0;
    for (u64 first = 0; first < n_blocks && !rc; first += per, k ^= 1) {
        rc = drain(k);
        if (rc) break;
        const u64 cnt = std::min<u64>(per, n_blocks - first);
        for (u64 i = 0; i < cnt && !rc; ++i) {
            const u64 a0 = offsets[first + i], a1 = offsets[first + i + 1];
            if (a1 < a0 || a1 - a0 > kMaxIn) { set_error("block %llu is not a piece of at most %u bytes", (unsigned long long)(first + i), kMaxIn); rc = -EINVAL; }
        }
        if (rc) break;
        const u64 b0 = offsets[first], b1 = offsets[first + cnt];
        memcpy(bt[k]->h_in, in + b0, b1 - b0);
        for (u64 i = 0; i <= cnt; ++i) bt[k]->h_off[i] = offsets[first + i] - b0;
        rc = mgx_bgzf_batch_submit(c, bt[k], (u32)cnt);
        if (!rc) fl[k] = {first, cnt};
    }
    if (!rc) rc = drain(k);
    if (!rc) rc = drain(k ^ 1);
    for (int i = 0; i < 2; ++i) mgx_bgzf_batch_destroy(c, bt[i]);
    return rc;
}

// ---- record store ---------------------------------------------------------------------------------------------------
struct mgx_bgzf_store {
    mgx_bgzf* ctx = nullptr;
    std::mutex mu;
    std::vector<u8*> chunks;
    u64 cur_off = 0, cur_cap = 0, total = 0;
    // put() stages through pinned buffers of its own (a copy from pageable memory is slow and serialises the callers)
    // and copies on streams of its own, never on the null stream (which would wait for every blocking stream of the process)
    static constexpr int kStage = 24;
    static constexpr size_t kStageBytes = 4u << 20;
    hipStream_t copy[kStage] = {};
    u8* stage[kStage] = {};                    // pinned, allocated by the first put that takes the slot (in parallel, off create's path)
    std::mutex stage_mu; std::condition_variable stage_cv;
    std::vector<int> stage_free;
};

int mgx_bgzf_store_create(mgx_bgzf_t* c, mgx_bgzf_store_t** out) {
    if (!c || !out) { set_error("NULL argument"); return -EINVAL; }
    *out = nullptr;
    mgx_bgzf_store* st = new (std::nothrow) mgx_bgzf_store;
    if (!st) { set_error("out of memory"); return -ENOMEM; }
    st->ctx = c;
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) {
        set_error("record store: %s", hipGetErrorString(e));
        mgx_bgzf_store_destroy(st);
        return -EIO;
    }
    // a slot's stream and pinned buffer are made by the first put that takes the slot: in parallel on the callers'
    // threads instead of 24 stream creations and 96 MB of page pinning in front of the first put
    for (int i = 0; i < mgx_bgzf_store::kStage; ++i) st->stage_free.push_back(mgx_bgzf_store::kStage - 1 - i);
    *out = st;
    return 0;
}

// Allocates HBM for about `bytes` more record bytes now (1 GB pieces), so that the puts of the first slices do not wait
// for hipMalloc one after the other.
int mgx_bgzf_store_reserve(mgx_bgzf_store_t* st, uint64_t bytes) {
    if (!st) { set_error("NULL argument"); return -EINVAL; }
    HIP_TRY(hipSetDevice(st->ctx->device));
    std::lock_guard<std::mutex> g(st->mu);
    if (!st->chunks.empty() || bytes == 0) return 0;
    const u64 cap = std::min<u64>(std::max<u64>((bytes + 15) & ~15ull, 64ull << 20), 256ull << 20);      // a first piece that is quick to map
    u8* p = nullptr;
    if (hipMalloc((void**)&p, cap) != hipSuccess) { set_error("hipMalloc of %llu bytes for the record store failed", (unsigned long long)cap); return -ENOMEM; }
    st->chunks.push_back(p); st->cur_off = 0; st->cur_cap = cap;
    return 0;
}

void mgx_bgzf_store_destroy(mgx_bgzf_store_t* st) {
    if (!st) return;
    (void)hipSetDevice(st->ctx->device);
    for (u8* p : st->chunks) (void)hipFree(p);
    for (auto& sc : st->copy) if (sc) (void)hipStreamDestroy(sc);
    for (auto& p : st->stage) if (p) (void)hipHostFree(p);
    delete st;
}

int mgx_bgzf_store_put(mgx_bgzf_store_t* st, const uint8_t* bytes, uint64_t n, uint64_t* addr) {
    if (!st || !addr || (n && !bytes)) { set_error("NULL argument"); return -EINVAL; }
    if (n == 0) { *addr = 0; return 0; }           // nothing to store (a slice of blank lines); no chunk is touched
    HIP_TRY(hipSetDevice(st->ctx->device));
    u8* dst;
    {
        std::lock_guard<std::mutex> g(st->mu);
        const u64 need = (n + 15) & ~15ull;
        if (st->cur_off + need > st->cur_cap) {
            const u64 cap = std::max<u64>(need, 1ull << 30);          // HBM in 1 GB pieces; a put never straddles two
            u8* p = nullptr;
            if (hipMalloc((void**)&p, cap) != hipSuccess) { set_error("hipMalloc of %llu bytes for the record store failed (%llu bytes stored)", (unsigned long long)cap, (unsigned long long)st->total); return -ENOMEM; }
            st->chunks.push_back(p); st->cur_off = 0; st->cur_cap = cap;
        }
        dst = st->chunks.back() + st->cur_off;
        st->cur_off += need; st->total += n;
    }
    for (u64 done = 0; done < n;) {
        int k;
        {
            std::unique_lock<std::mutex> lk(st->stage_mu);
            st->stage_cv.wait(lk, [&] { return !st->stage_free.empty(); });
            k = st->stage_free.back(); st->stage_free.pop_back();
        }
        const u64 piece = std::min<u64>(mgx_bgzf_store::kStageBytes, n - done);
        hipError_t e = hipSuccess;
        if (!st->copy[k]) e = hipStreamCreateWithFlags(&st->copy[k], hipStreamNonBlocking);                                     // this slot's first use
        if (e == hipSuccess && !st->stage[k]) e = hipHostMalloc((void**)&st->stage[k], mgx_bgzf_store::kStageBytes, hipHostMallocDefault);
        if (e == hipSuccess) {
            memcpy(st->stage[k], bytes + done, piece);
            e = hipMemcpyAsync(dst + done, st->stage[k], piece, hipMemcpyHostToDevice, st->copy[k]);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st->copy[k]);
        { std::lock_guard<std::mutex> g(st->stage_mu); st->stage_free.push_back(k); }
        st->stage_cv.notify_one();
        if (e != hipSuccess) { set_error("copy of %llu bytes to the record store: %s", (unsigned long long)piece, hipGetErrorString(e)); return -EIO; }
        done += piece;
    }
    *addr = (uint64_t)(uintptr_t)dst;
    return 0;
}

int mgx_bgzf_store_emit(mgx_bgzf_store_t* st, uint64_t n, const uint32_t* order, const uint8_t* dup, const uint64_t* addr, const uint32_t* len,
                        mgx_bgzf_sink_t sink, void* user, uint64_t* uoff_out) {
    if (!st || !sink || !uoff_out || (n && (!order || !dup || !addr || !len))) { set_error("NULL argument"); return -EINVAL; }
    mgx_bgzf* c = st->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const bool trace = getenv("MGX_BGZF_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto now = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); };
    uoff_out[0] = 0;
    if (n == 0) return 0;
    if (n > 0xFFFFFFF0ull) { set_error("more than 2^32 records"); return -E2BIG; }
    hipStream_t s = c->stream;
    u32 *d_order = nullptr, *d_len = nullptr; u8* d_dup = nullptr; u64 *d_addr = nullptr, *d_uoff = nullptr, *d_tiles = nullptr;
    const u32 n_tiles = (u32)((n + kScanTile - 1) / kScanTile);
    int rc = 0;
    mgx_bgzf_batch_t* bt[3] = {nullptr, nullptr, nullptr};
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(s);
        (void)hipFree(d_order); (void)hipFree(d_len); (void)hipFree(d_dup); (void)hipFree(d_addr); (void)hipFree(d_uoff); (void)hipFree(d_tiles);
        for (auto* b : bt) if (b) mgx_bgzf_batch_destroy(c, b);
    };
#define STORE_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); cleanup(); return -EIO; } } while (0)
    STORE_TRY(hipMalloc((void**)&d_order, n * sizeof(u32)));
    STORE_TRY(hipMalloc((void**)&d_len, n * sizeof(u32)));
    STORE_TRY(hipMalloc((void**)&d_dup, n));
    STORE_TRY(hipMalloc((void**)&d_addr, n * sizeof(u64)));
    STORE_TRY(hipMalloc((void**)&d_uoff, (n + 1) * sizeof(u64)));
    STORE_TRY(hipMalloc((void**)&d_tiles, ((size_t)n_tiles + 1) * sizeof(u64)));
    STORE_TRY(hipMemcpyAsync(d_order, order, n * sizeof(u32), hipMemcpyHostToDevice, s));
    STORE_TRY(hipMemcpyAsync(d_len, len, n * sizeof(u32), hipMemcpyHostToDevice, s));
    STORE_TRY(hipMemcpyAsync(d_dup, dup, n, hipMemcpyHostToDevice, s));
    STORE_TRY(hipMemcpyAsync(d_addr, addr, n * sizeof(u64), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_store_tile_sums, dim3(n_tiles), dim3(256), 0, s, d_order, d_len, n, d_tiles);
    hipLaunchKernelGGL(k_store_scan_tiles, dim3(1), dim3(1024), 0, s, d_tiles, n_tiles, d_uoff + n);
    hipLaunchKernelGGL(k_store_offsets, dim3(n_tiles), dim3(256), 0, s, d_order, d_len, n, d_tiles, d_uoff);
    STORE_TRY(hipGetLastError());
    STORE_TRY(hipMemcpyAsync(uoff_out, d_uoff, (n + 1) * sizeof(u64), hipMemcpyDeviceToHost, s));
    STORE_TRY(hipStreamSynchronize(s));
    const u64 total = uoff_out[n];
    double t_wait = 0, t_sink = 0;
    if (trace) fprintf(stderr, "  store_emit: arrays on the device, offsets scanned and back: %.3f s\n", now());
    constexpr u32 kPer = 1024;                                    // blocks per batch (67 MB of the stream)
    const u64 win = (u64)kPer * kMaxIn;
    const u64 n_win = (total + win - 1) / win;
    for (int i = 0; i < 3 && !rc && (u64)i < n_win; ++i) rc = batch_create(c, win, kPer, false, &bt[i]);
    if (rc) { cleanup(); return rc; }
    if (trace) fprintf(stderr, "  store_emit: batches allocated: %.3f s\n", now());
    auto collect = [&](u64 k) -> int {
        const uint8_t* o; const uint64_t* oo;
        mgx_bgzf_batch_t* b = bt[k % 3];
        const double t0 = now();
        const int r = mgx_bgzf_batch_wait(c, b, &o, &oo);
        if (r) return r;
        const double t1 = now();
        const int sr = sink(user, o, oo[b->n_blocks], b->n_blocks, oo);
        t_wait += t1 - t0; t_sink += now() - t1;
        if (sr) { set_error("the sink returned %d", sr); return -EIO; }
        return 0;
    };
    u64 q_lo = 0;
    for (u64 k = 0; k < n_win && !rc; ++k) {
        if (k >= 3) rc = collect(k - 3);
        if (rc) break;
        mgx_bgzf_batch_t* b = bt[k % 3];
        const u64 w0 = k * win, wb = std::min(win, total - w0);
        // records that touch [w0, w0 + wb): from the one holding byte w0 to the last one starting before the window's end
        while (q_lo + 1 <= n && uoff_out[q_lo + 1] <= w0) ++q_lo;
        const u64 q_hi = (u64)(std::lower_bound(uoff_out + q_lo, uoff_out + n, w0 + wb) - uoff_out);
        const u32 nb = (u32)((wb + kMaxIn - 1) / kMaxIn);
        for (u32 i = 0; i <= nb; ++i) b->h_off[i] = std::min<u64>((u64)i * kMaxIn, wb);
        if (q_hi > q_lo)
            hipLaunchKernelGGL(k_store_gather, dim3((u32)((q_hi - q_lo + 3) / 4)), dim3(256), 0, s, d_addr, d_len, d_order, d_dup, d_uoff, q_lo, q_hi, w0, wb, b->d_in);
        rc = batch_submit(c, b, nb, true);
    }
    for (u64 k = n_win >= 3 ? n_win - 3 : 0; k < n_win && !rc; ++k) rc = collect(k);
    if (trace) fprintf(stderr, "  store_emit: %llu windows done: %.3f s (waiting for the device %.3f s, in the sink %.3f s)\n", (unsigned long long)n_win, now(), t_wait, t_sink);
#undef STORE_TRY
    cleanup();
    return rc;
}

int mgx_bgzf_stats(mgx_bgzf_t* c, mgx_bgzf_stats_t* out) {
    if (!c || !out) { set_error("NULL argument"); return -EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    u32 ns = 0;
    if (c->d_n_stored) HIP_TRY(hipMemcpy(&ns, c->d_n_stored, sizeof ns, hipMemcpyDeviceToHost));
    out->n_blocks = c->n_blocks; out->bytes_in = c->bytes_in; out->bytes_out = c->bytes_out; out->n_stored = ns; out->ms_kernels = c->ms_kernels; out->ms_pack = c->ms_pack;
    return 0;
}

}  // extern "C"
