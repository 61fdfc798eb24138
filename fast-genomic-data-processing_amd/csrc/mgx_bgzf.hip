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
#include <cerrno>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
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

constexpr int kNT = 256;                       // threads per workgroup = positions per match window
constexpr int kHashBits = 12;
constexpr u32 kMaxIn = MGX_BGZF_MAX_BLOCK_IN;
constexpr u32 kSlot = 0x10000;                 // bytes of device scratch per finished block
constexpr u32 kSlotSkew = 2;                   // a block starts at slot + 2: its payload (18 bytes in) is 4-byte aligned
constexpr u32 kPad = 320;                      // zero bytes after the input in LDS (match extension reads ahead)
constexpr u32 kCrcPoly = 0xEDB88320u;
constexpr int kNumLL = 286, kNumD = 30, kNumCL = 19;

enum { V_OVER = 0, V_NLIT, V_NDIST, V_NCLSYM, V_HDRBITS, V_STORED, V_CRC, V_N };

struct __attribute__((aligned(16))) Lds {
    u32 buf[(0x10000 + 512) / 4];              // the block's bytes (at the source's alignment), later the output words
    u32 head[1 << kHashBits];                  // hash -> last position + 1
    u32 crc_tab[256];
    u32 x2n[32];                               // x^(2^k) mod P
    u32 f_ll[288], f_d[32], f_cl[20];          // symbol counts
    u16 c_ll[288], c_d[32], c_cl[20];          // codes, bit-reversed for LSB-first output
    u8 l_ll[288], l_d[32], l_cl[20];           // code lengths
    u16 sorted[288];
    u32 wl[288], wi[288];                      // weights: leaves (sorted), internal nodes (in creation order)
    u16 parent[576];
    u8 depth[576];
    u32 bl_count[16], next_code[16];
    u32 scan[kNT];
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
    u32* scratch;          // [gridDim.x][65536]: match table, then the tokens
    u32* n_stored;         // counter
    u32 lazy;
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

__device__ __forceinline__ u32 match_len(const u8* a, const u8* b, u32 maxlen) {
    u32 len = 0;
    while (len + 4 <= maxlen) {
        const u32 x = load32(a + len) ^ load32(b + len);
        if (x) return len + (__builtin_ctz(x) >> 3);
        len += 4;
    }
    while (len < maxlen && a[len] == b[len]) ++len;
    return len;
}

// Length-limited Huffman code of freq[0, N): lengths and (bit-reversed) canonical codes.  Called by the whole workgroup.
// At least two symbols get a code (inflate accepts no incomplete literal/length or code-length code).
__device__ void huff_build(Lds& L, u32* freq, const int N, const int limit, u8* len, u16* code) {
    const int tid = (int)threadIdx.x;
    __syncthreads();
    if (tid == 0) {
        int used = 0;
        for (int s = 0; s < N; ++s) used += freq[s] != 0;
        if (used == 0) { freq[0] = 1; freq[1] = 1; }
        else if (used == 1) freq[freq[0] ? 1 : 0] = 1;
    }
    for (;;) {
        __syncthreads();
        for (int s = tid; s < N; s += kNT) {
            len[s] = 0;
            const u32 f = freq[s];
            if (f) {
                int r = 0;
                for (int j = 0; j < N; ++j) { const u32 g = freq[j]; r += (g != 0) & ((g < f) | ((g == f) & (j < s))); }
                L.sorted[r] = (u16)s;
            }
        }
        __syncthreads();
        if (tid == 0) {
            int n = 0;
            for (int s = 0; s < N; ++s) n += freq[s] != 0;
            for (int i = 0; i < n; ++i) L.wl[i] = freq[L.sorted[i]];
            int i = 0, j = 0;
            for (int k = 0; k < n - 1; ++k) {            // two-queue merge: leaves in weight order, internal nodes in creation order
                u32 w2 = 0;
                for (int t = 0; t < 2; ++t) {
                    int id;
                    if (i < n && (j >= k || L.wl[i] <= L.wi[j])) { id = i; w2 += L.wl[i]; ++i; }
                    else { id = n + j; w2 += L.wi[j]; ++j; }
                    L.parent[id] = (u16)(n + k);
                }
                L.wi[k] = w2;
            }
            L.depth[2 * n - 2] = 0;
            int maxd = 0;
            for (int id = 2 * n - 3; id >= 0; --id) {
                const int d = L.depth[L.parent[id]] + 1;
                L.depth[id] = (u8)d;
                if (id < n && d > maxd) maxd = d;
            }
            if (maxd <= limit) for (int q = 0; q < n; ++q) len[L.sorted[q]] = L.depth[q];
            L.vars[V_OVER] = maxd > limit;
        }
        __syncthreads();
        if (!L.vars[V_OVER]) break;
        for (int s = tid; s < N; s += kNT) { const u32 f = freq[s]; if (f) freq[s] = (f + 1) >> 1; }     // flatter, still >= 1
    }
    if (tid < 16) {
        u32 c = 0;
        for (int s = 0; s < N; ++s) c += (len[s] == tid);
        L.bl_count[tid] = tid ? c : 0;
    }
    __syncthreads();
    if (tid == 0) {
        u32 c = 0;
        L.next_code[0] = 0;
        for (int bits = 1; bits <= 15; ++bits) { c = (c + L.bl_count[bits - 1]) << 1; L.next_code[bits] = c; }
    }
    __syncthreads();
    for (int s = tid; s < N; s += kNT) {
        const u32 l = len[s];
        u32 c = 0;
        if (l) {
            c = L.next_code[l];
            for (int j = 0; j < s; ++j) c += (len[j] == l);
            c = __brev(c) >> (32 - l);
        }
        code[s] = (u16)c;
    }
    __syncthreads();
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

__global__ __launch_bounds__(kNT) void k_bgzf_deflate(DeflateArgs a) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    Lds& L = *reinterpret_cast<Lds*>(smem);
    const u32 tid = threadIdx.x;
    u32* const mat = a.scratch + (size_t)blockIdx.x * 65536u;

    for (u32 i = tid; i < 256; i += kNT) {
        u32 c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
        L.crc_tab[i] = c;
    }
    if (tid == 0) {
        u32 p = 0x40000000u;                   // x^1
        L.x2n[0] = p;
        for (int k = 1; k < 32; ++k) { p = multmodp(p, p); L.x2n[k] = p; }
    }
    __syncthreads();

    for (u32 blk = blockIdx.x; blk < a.n_blocks; blk += gridDim.x) {
        const u64 o0 = a.off[blk];
        const u32 n = (u32)(a.off[blk + 1] - o0);
        const u32 mis = (u32)(o0 & 3u);
        u8* const in = reinterpret_cast<u8*>(L.buf) + mis;           // the LDS copy keeps the source's word alignment
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
        __syncthreads();

        // ---- 1. match search
        for (u32 base = 0; base < n; base += kNT) {
            const u32 p = base + tid;
            const bool hashed = p + 4 <= n;
            u32 h = 0, cand = 0;
            if (hashed) { h = (load32(in + p) * 2654435761u) >> (32 - kHashBits); cand = L.head[h]; }
            __syncthreads();
            if (hashed) atomicMax(&L.head[h], p + 1);
            if (p < n) {
                const u32 maxlen = min(258u, n - p);
                u32 best_len = 0, best_dist = 0;
                if (cand) {
                    const u32 q = cand - 1, dist = p - q;
                    if (dist <= 32768u) {
                        const u32 len = match_len(in + q, in + p, maxlen);
                        if (len >= 4) { best_len = len; best_dist = dist; }
                    }
                }
                if (p >= 1 && maxlen >= 3 && in[p - 1] == in[p]) {
                    const u32 len = match_len(in + p - 1, in + p, maxlen);
                    if (len >= 3 && len >= best_len) { best_len = len; best_dist = 1; }
                }
                mat[p] = best_len ? (best_len << 16 | best_dist) : 0u;
            }
            __syncthreads();
        }

        // ---- 2. parse: thread t owns bytes [lo, hi)
        const u32 chunk = (n + kNT - 1) / kNT;
        const u32 lo = min(n, tid * chunk), hi = min(n, lo + chunk);
        u32 n_tok = 0;
        {
            u32 p = lo;
            while (p < hi) {
                const u32 m = mat[p];
                u32 len = min(m >> 16, hi - p), dist = m & 0xffffu;
                if (len < 3) len = 0;
                if (len && a.lazy && p + 1 < hi) {
                    const u32 len2 = min(mat[p + 1] >> 16, hi - p - 1);
                    if (len2 > len) len = 0;                          // a longer match starts one byte on: literal now
                }
                u32 tok;
                if (len) {
                    u32 ls, leb, lev, ds, deb, dev;
                    length_code(len, &ls, &leb, &lev);
                    dist_code(dist, &ds, &deb, &dev);
                    atomicAdd(&L.f_ll[ls], 1u);
                    atomicAdd(&L.f_d[ds], 1u);
                    tok = 0x80000000u | (len - 3) << 16 | (dist - 1);
                    p += len;
                } else {
                    tok = in[p];
                    atomicAdd(&L.f_ll[tok], 1u);
                    p += 1;
                }
                mat[lo + n_tok++] = tok;        // in place: token k of this thread lies at or before the byte it starts at
            }
        }
        // ---- 5. CRC-32 of the piece, shifted to the end of the block
        u32 crc_part;
        {
            u32 c = tid == 0 ? 0xFFFFFFFFu : 0u;
            for (u32 p = lo; p < hi; ++p) c = L.crc_tab[(c ^ in[p]) & 0xffu] ^ (c >> 8);
            u32 e = 8u * (n - hi), xp = 0x80000000u;
            for (int k = 0; e; ++k, e >>= 1) if (e & 1u) xp = multmodp(L.x2n[k], xp);
            crc_part = multmodp(xp, c);
        }
        L.scan[tid] = crc_part;
        if (tid == 0) L.f_ll[256] = 1;             // end of block
        __syncthreads();
        if (tid == 0) {
            u32 c = 0;
            for (int i = 0; i < kNT; ++i) c ^= L.scan[i];
            L.vars[V_CRC] = c ^ 0xFFFFFFFFu;
        }

        // ---- 3. codes
        huff_build(L, L.f_ll, kNumLL, 15, L.l_ll, L.c_ll);
        huff_build(L, L.f_d, kNumD, 15, L.l_d, L.c_d);
        if (tid == 0) {
            int nlit = kNumLL, ndist = kNumD;
            while (nlit > 257 && L.l_ll[nlit - 1] == 0) --nlit;
            while (ndist > 1 && L.l_d[ndist - 1] == 0) --ndist;
            L.vars[V_NLIT] = nlit; L.vars[V_NDIST] = ndist;
            // run-length code of the nlit + ndist code lengths (symbols 16 / 17 / 18)
            const int total = nlit + ndist;
            auto length_at = [&](int i) -> u32 { return i < nlit ? L.l_ll[i] : L.l_d[i - nlit]; };
            int ns = 0;
            auto emit = [&](u32 sym, u32 extra) { L.clsym[ns++] = (u16)(sym | extra << 8); L.f_cl[sym] += 1; };
            for (int i = 0; i < total;) {
                const u32 v = length_at(i);
                int r = 1;
                while (i + r < total && length_at(i + r) == v) ++r;
                i += r;
                if (v == 0) {
                    while (r > 0) {
                        if (r >= 11) { const int t = r < 138 ? r : 138; emit(18, (u32)(t - 11)); r -= t; }
                        else if (r >= 3) { emit(17, (u32)(r - 3)); r = 0; }
                        else { emit(0, 0); --r; }
                    }
                } else {
                    emit(v, 0); --r;
                    while (r > 0) {
                        if (r >= 3) { const int t = r < 6 ? r : 6; emit(16, (u32)(t - 3)); r -= t; }
                        else { emit(v, 0); --r; }
                    }
                }
            }
            L.vars[V_NCLSYM] = ns;
        }
        huff_build(L, L.f_cl, kNumCL, 7, L.l_cl, L.c_cl);
        if (tid == 0) {
            const u8 order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            for (int i = 0; i < 192; ++i) L.hdr[i] = 0;
            u32 word = 0, nbits = 0; u64 acc = 0;
            auto put = [&](u32 v, u32 nb) {
                acc |= (u64)v << nbits; nbits += nb;
                if (nbits >= 32) { L.hdr[word++] = (u32)acc; acc >>= 32; nbits -= 32; }
            };
            int ncl = 19;
            while (ncl > 4 && L.l_cl[order[ncl - 1]] == 0) --ncl;
            put(1u | 2u << 1, 3);                                   // BFINAL = 1, BTYPE = 10 (dynamic)
            put(L.vars[V_NLIT] - 257, 5); put(L.vars[V_NDIST] - 1, 5); put((u32)ncl - 4, 4);
            for (int i = 0; i < ncl; ++i) put(L.l_cl[order[i]], 3);
            const int ns = (int)L.vars[V_NCLSYM];
            for (int i = 0; i < ns; ++i) {
                const u32 sym = L.clsym[i] & 0xffu, extra = L.clsym[i] >> 8;
                put(L.c_cl[sym], L.l_cl[sym]);
                if (sym == 16) put(extra, 2); else if (sym == 17) put(extra, 3); else if (sym == 18) put(extra, 7);
            }
            L.vars[V_HDRBITS] = word * 32 + nbits;
            if (nbits) L.hdr[word] = (u32)acc;
        }
        __syncthreads();

        // ---- 4. measure, scan, emit
        u32 my_bits = 0;
        for (u32 k = 0; k < n_tok; ++k) {
            const u32 tok = mat[lo + k];
            if (tok & 0x80000000u) {
                u32 ls, leb, lev, ds, deb, dev;
                length_code(((tok >> 16) & 0xffu) + 3, &ls, &leb, &lev);
                dist_code((tok & 0xffffu) + 1, &ds, &deb, &dev);
                my_bits += L.l_ll[ls] + leb + L.l_d[ds] + deb;
            } else {
                my_bits += L.l_ll[tok];
            }
        }
        L.scan[tid] = my_bits;
        __syncthreads();
        for (u32 off = 1; off < kNT; off <<= 1) {
            const u32 v = tid >= off ? L.scan[tid - off] : 0;
            __syncthreads();
            L.scan[tid] += v;
            __syncthreads();
        }
        const u32 hdr_bits = L.vars[V_HDRBITS];
        const u32 total_bits = hdr_bits + L.scan[kNT - 1] + L.l_ll[256];
        const u32 my_start = hdr_bits + L.scan[tid] - my_bits;
        u32 payload = (total_bits + 7) >> 3;
        const bool stored = payload >= n + 5 || n == 0;
        u8* const slot = a.slots + (size_t)blk * kSlot + kSlotSkew;
        u8* const pay = slot + 18;
        __syncthreads();                                    // every reader of the input bytes and of scan[] is done
        if (!stored) {
            const u32 nw = (payload + 3) >> 2;
            for (u32 w = tid; w < nw; w += kNT) L.buf[w] = 0;
            __syncthreads();
            for (u32 w = tid; w < ((hdr_bits + 31) >> 5); w += kNT) atomicOr(&L.buf[w], L.hdr[w]);
            BitSink s;
            s.start(L.buf, my_start);
            for (u32 k = 0; k < n_tok; ++k) {
                const u32 tok = mat[lo + k];
                if (tok & 0x80000000u) {
                    u32 ls, leb, lev, ds, deb, dev;
                    length_code(((tok >> 16) & 0xffu) + 3, &ls, &leb, &lev);
                    dist_code((tok & 0xffffu) + 1, &ds, &deb, &dev);
                    s.put(L.c_ll[ls], L.l_ll[ls]);
                    if (leb) s.put(lev, leb);
                    s.put(L.c_d[ds], L.l_d[ds]);
                    if (deb) s.put(dev, deb);
                } else {
                    s.put(L.c_ll[tok], L.l_ll[tok]);
                }
            }
            if (tid == kNT - 1) s.put(L.c_ll[256], L.l_ll[256]);
            s.finish();
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

// slot -> its place in the packed output
__global__ __launch_bounds__(256) void k_bgzf_pack(const u8* slots, const u32* sizes, const u64* out_off, u8* out) {
    const u32 blk = blockIdx.x;
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

}  // namespace

struct mgx_bgzf {
    int device = 0;
    int n_cu = 0;
    hipStream_t stream = nullptr;              // all batches of a context run in order on one stream
    u32* d_scratch = nullptr; u32 grid = 0;
    u32* d_n_stored = nullptr;
    u32 lazy = 1;
    u64 n_blocks = 0, bytes_in = 0, bytes_out = 0;
    float ms_kernels = 0;
};

struct mgx_bgzf_batch {
    u64 in_cap = 0; u32 max_blocks = 0;
    u8* h_in = nullptr; u64* h_off = nullptr;          // pinned, filled by the caller
    u8* h_out = nullptr; u64* h_out_off = nullptr;     // pinned, results
    u8* d_in = nullptr; u64* d_off = nullptr; u8* d_slots = nullptr; u32* d_sizes = nullptr; u64* d_out_off = nullptr; u8* d_out = nullptr;
    u64 out_cap = 0;
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_off = nullptr;
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
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->grid = (u32)c->n_cu;                       // ~94 KB of LDS per workgroup: one per CU
    if (const char* e = getenv("MGX_BGZF_GRID")) { const int v = atoi(e); if (v > 0) c->grid = (u32)v; }
    HIP_TRY(hipMalloc((void**)&c->d_scratch, (size_t)c->grid * 65536u * sizeof(u32)));
    HIP_TRY(hipMalloc((void**)&c->d_n_stored, sizeof(u32)));
    HIP_TRY(hipMemset(c->d_n_stored, 0, sizeof(u32)));
    HIP_TRY(hipFuncSetAttribute((const void*)k_bgzf_deflate, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds)));
    *out = c.release();
    return 0;
}

void mgx_bgzf_destroy(mgx_bgzf_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    (void)hipFree(c->d_scratch);
    (void)hipFree(c->d_n_stored);
    delete c;
}

void mgx_bgzf_batch_destroy(mgx_bgzf_t* c, mgx_bgzf_batch_t* b) {
    if (!b) return;
    if (c) { (void)hipSetDevice(c->device); (void)hipStreamSynchronize(c->stream); }
    (void)hipHostFree(b->h_in); (void)hipHostFree(b->h_off); (void)hipHostFree(b->h_out); (void)hipHostFree(b->h_out_off);
    (void)hipFree(b->d_in); (void)hipFree(b->d_off); (void)hipFree(b->d_slots); (void)hipFree(b->d_sizes); (void)hipFree(b->d_out_off); (void)hipFree(b->d_out);
    if (b->ev_k0) (void)hipEventDestroy(b->ev_k0);
    if (b->ev_k1) (void)hipEventDestroy(b->ev_k1);
    if (b->ev_off) (void)hipEventDestroy(b->ev_off);
    delete b;
}

int mgx_bgzf_batch_create(mgx_bgzf_t* c, uint64_t in_capacity, uint32_t max_blocks, mgx_bgzf_batch_t** out) {
    if (!c || !out) { set_error("NULL argument"); return -EINVAL; }
    *out = nullptr;
    if (max_blocks == 0) { set_error("max_blocks is 0"); return -EINVAL; }
    in_capacity = std::min<u64>(in_capacity, (u64)max_blocks * kMaxIn);
    HIP_TRY(hipSetDevice(c->device));
    mgx_bgzf_batch* b = new (std::nothrow) mgx_bgzf_batch;
    if (!b) { set_error("out of memory"); return -ENOMEM; }
    b->in_cap = in_capacity; b->max_blocks = max_blocks;
    b->out_cap = mgx_bgzf_bound(in_capacity, max_blocks);
    auto fail = [&](const char* what) { set_error("%s failed for a batch of %llu bytes / %u blocks", what, (unsigned long long)in_capacity, max_blocks); mgx_bgzf_batch_destroy(c, b); return -ENOMEM; };
    if (hipHostMalloc((void**)&b->h_in, in_capacity + 8, hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc");
    if (hipHostMalloc((void**)&b->h_off, ((size_t)max_blocks + 1) * sizeof(u64), hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc");
    if (hipHostMalloc((void**)&b->h_out, b->out_cap, hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc");
    if (hipHostMalloc((void**)&b->h_out_off, ((size_t)max_blocks + 1) * sizeof(u64), hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc");
    if (hipMalloc((void**)&b->d_in, in_capacity + 8) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&b->d_off, ((size_t)max_blocks + 1) * sizeof(u64)) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&b->d_slots, (size_t)max_blocks * kSlot) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&b->d_sizes, (size_t)max_blocks * sizeof(u32)) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&b->d_out_off, ((size_t)max_blocks + 1) * sizeof(u64)) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&b->d_out, b->out_cap) != hipSuccess) return fail("hipMalloc");
    if (hipEventCreate(&b->ev_k0) != hipSuccess || hipEventCreate(&b->ev_k1) != hipSuccess ||
        hipEventCreateWithFlags(&b->ev_off, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate");
    b->h_off[0] = 0;
    *out = b;
    return 0;
}

uint8_t* mgx_bgzf_batch_input(mgx_bgzf_batch_t* b) { return b ? b->h_in : nullptr; }
uint64_t* mgx_bgzf_batch_offsets(mgx_bgzf_batch_t* b) { return b ? b->h_off : nullptr; }

int mgx_bgzf_batch_submit(mgx_bgzf_t* c, mgx_bgzf_batch_t* b, uint32_t n_blocks) {
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
    HIP_TRY(hipMemcpyAsync(b->d_in, b->h_in, n_in, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b->d_off, b->h_off, ((size_t)n_blocks + 1) * sizeof(u64), hipMemcpyHostToDevice, s));
    DeflateArgs a{};
    a.in = b->d_in; a.off = b->d_off; a.n_blocks = n_blocks; a.slots = b->d_slots; a.sizes = b->d_sizes;
    a.scratch = c->d_scratch; a.n_stored = c->d_n_stored; a.lazy = c->lazy;
    HIP_TRY(hipEventRecord(b->ev_k0, s));
    hipLaunchKernelGGL(k_bgzf_deflate, dim3(std::min(c->grid, n_blocks)), dim3(kNT), sizeof(Lds), s, a);
    hipLaunchKernelGGL(k_bgzf_offsets, dim3(1), dim3(256), 0, s, b->d_sizes, n_blocks, b->d_out_off);
    hipLaunchKernelGGL(k_bgzf_pack, dim3(n_blocks), dim3(256), 0, s, b->d_slots, b->d_sizes, b->d_out_off, b->d_out);
    HIP_TRY(hipEventRecord(b->ev_k1, s));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(b->h_out_off, b->d_out_off, ((size_t)n_blocks + 1) * sizeof(u64), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(b->ev_off, s));
    return 0;
}

int mgx_bgzf_batch_wait(mgx_bgzf_t* c, mgx_bgzf_batch_t* b, const uint8_t** out, const uint64_t** out_offsets) {
    if (!c || !b || !out || !out_offsets) { set_error("NULL argument"); return -EINVAL; }
    if (!b->submitted) { set_error("batch was not submitted"); return -EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    *out = b->h_out; *out_offsets = b->h_out_off;
    b->submitted = false;
    if (b->n_blocks == 0) return 0;
    HIP_TRY(hipEventSynchronize(b->ev_off));
    const u64 total = b->h_out_off[b->n_blocks];
    if (total > b->out_cap) { set_error("internal: %llu output bytes exceed the bound %llu", (unsigned long long)total, (unsigned long long)b->out_cap); return -EIO; }
    // the packed blocks: exactly their bytes (the copy is ordered behind later batches' kernels on the stream; with
    // two or three batches in flight the link is busy either way)
    HIP_TRY(hipMemcpyAsync(b->h_out, b->d_out, total, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0;
    if (hipEventElapsedTime(&ms, b->ev_k0, b->ev_k1) == hipSuccess) c->ms_kernels = ms;
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

int mgx_bgzf_stats(mgx_bgzf_t* c, mgx_bgzf_stats_t* out) {
    if (!c || !out) { set_error("NULL argument"); return -EINVAL; }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    u32 ns = 0;
    HIP_TRY(hipMemcpy(&ns, c->d_n_stored, sizeof ns, hipMemcpyDeviceToHost));
    out->n_blocks = c->n_blocks; out->bytes_in = c->bytes_in; out->bytes_out = c->bytes_out; out->n_stored = ns; out->ms_kernels = c->ms_kernels;
    return 0;
}

}  // extern "C"
