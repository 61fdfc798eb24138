// tools/sort_microbench.hip -- probes for the radix scatter pass (development aid, not in libmgx.so):
//   copy      : u64 stream copy, the HBM read+write ceiling for an 8-byte-key pass
//   segscatter: every 4096-key tile is read contiguously and written as 256 runs of 16 keys (128 B),
//               run d of tile t going to d*(n/256) + t*16 -- the store pattern of an LSD-8 pass
//               over uniformly distributed digits, without any ranking work
//   matchA/B  : the two forms of the in-wavefront digit match, VALU only
//   hipcc -O3 --offload-arch=gfx950 tools/sort_microbench.hip -o tools/bin/sort_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
typedef uint32_t u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_copy(const u64* __restrict__ in, u64* __restrict__ out, u32 n) {
    const u32 base = blockIdx.x * 4096;
    u64 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { const u32 i = base + k * 256 + threadIdx.x; v[k] = i < n ? in[i] : 0; }
#pragma unroll
    for (int k = 0; k < 16; ++k) { const u32 i = base + k * 256 + threadIdx.x; if (i < n) out[i] = v[k]; }
}

template <int RUN, bool XCD>   // RUN keys per run (16 = 128 B, 2 = 16 B ...); XCD: each XCD walks a contiguous tile range
__global__ __launch_bounds__(256) void k_segscatter(const u64* __restrict__ in, u64* __restrict__ out, u32 n, u32 n_tiles) {
    u32 t = blockIdx.x;
    if (XCD) { const u32 q = n_tiles >> 3, r = n_tiles & 7u, x = blockIdx.x & 7u; t = x * q + min(x, r) + (blockIdx.x >> 3); }
    const u32 base = t * 4096;
    constexpr int kRuns = 4096 / RUN;                 // runs per tile = number of buckets
    const u32 bucket_len = n / kRuns;                 // n is a multiple of 4096
    u64 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = in[base + k * 256 + threadIdx.x];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const u32 i = k * 256 + threadIdx.x;          // tile-local sorted slot
        const u32 d = i / RUN, j = i % RUN;
        out[(size_t)d * bucket_len + (size_t)t * RUN + j] = v[k];
    }
}

// ---- staged replica of the scatter pass: STAGE 0 = load + LDS exchange with a fixed permutation + store,
// 1 = + digit match / rank arithmetic (result discarded into the permutation), 2 = + LDS counter atomics
// and leader broadcast, 3 = the full kernel (block scan, real slots).  Offsets are synthetic (bucket d of
// tile t at d*bucket_len + t*16), which is the real store pattern for uniform digits.
template <int STAGE>
__global__ __launch_bounds__(256) void k_stage(const u64* __restrict__ kin, u64* __restrict__ kout, u32 n, int shift) {
    __shared__ u64 sbuf[4096];
    __shared__ u32 wcnt[4][256];
    __shared__ u32 tbase[256];
    __shared__ u32 sm[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 t = blockIdx.x, tile0 = t * 4096;
    const u32 bucket_len = n / 256;
    for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0;
    __syncthreads();
    u64 key[16]; u32 lrank[16]; u32 dig[16];
    const u64 lt = (1ull << lane) - 1;
#pragma unroll
    for (int k = 0; k < 16; ++k) key[k] = kin[tile0 + wave * 1024 + k * 64 + lane];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const u32 d = (u32)(key[k] >> shift) & 255u;
        dig[k] = d;
        u32 r = 0, old = 0;
        if constexpr (STAGE >= 1) {
            u64 peers = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; ++b) { const bool bit = (d >> b) & 1u; const u64 m = __ballot(bit); peers &= bit ? m : ~m; }
            r = __popcll(peers & lt);
            if constexpr (STAGE >= 2) {
                if (r == 0) old = atomicAdd(&wcnt[wave][d], (u32)__popcll(peers));
                old = __shfl(old, __ffsll((long long)peers) - 1, 64);
            }
        }
        lrank[k] = old + r;
    }
    __syncthreads();
    if constexpr (STAGE >= 3) {
        const u32 c0 = wcnt[0][tid], c1 = wcnt[1][tid], c2 = wcnt[2][tid], c3 = wcnt[3][tid];
        u32 v = c0 + c1 + c2 + c3, incl = v;
        for (int off = 1; off < 64; off <<= 1) { u32 x = __shfl_up(incl, off, 64); if (lane >= off) incl += x; }
        if (lane == 63) sm[wave] = incl;
        __syncthreads();
        u32 base = 0;
        for (int w = 0; w < 4; ++w) if (w < wave) base += sm[w];
        const u32 tb = base + incl - v;
        wcnt[0][tid] = tb; wcnt[1][tid] = tb + c0; wcnt[2][tid] = tb + c0 + c1; wcnt[3][tid] = tb + c0 + c1 + c2;
        tbase[tid] = tb;
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        u32 lpos;
        if constexpr (STAGE >= 3) lpos = wcnt[wave][dig[k]] + lrank[k];
        else lpos = ((wave * 1024 + k * 64 + lane) * 1031u + (lrank[k] & 0u)) & 4095u;     // fixed permutation (1031 is odd)
        sbuf[lpos] = key[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const u32 i = k * 256 + tid;
        const u64 kk = sbuf[i];
        u32 d, j;
        if constexpr (STAGE >= 3) { d = (u32)(kk >> shift) & 255u; j = (i - tbase[d]) & 31u; }
        else { d = i >> 4; j = i & 15u; }
        kout[(size_t)d * bucket_len + (size_t)t * 16 + j] = kk;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_match(u32* out, int iters) {
    u32 x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    u32 acc = 0;
    const u64 lt = (1ull << (threadIdx.x & 63)) - 1;
    const u32 lt_lo = (u32)lt, lt_hi = (u32)(lt >> 32);
    for (int it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        const u32 d = x >> 24;
        u32 plo, phi;
        if constexpr (MODE == 0) {
            u64 peers = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; ++b) { const bool bit = (d >> b) & 1u; const u64 m = __ballot(bit); peers &= bit ? m : ~m; }
            plo = (u32)peers; phi = (u32)(peers >> 32);
        } else {
            u32 dlo = 0, dhi = 0;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int sel = __builtin_amdgcn_sbfe((int)d, b, 1);
                const u64 m = __ballot(sel != 0);
                dlo |= (u32)m ^ (u32)sel; dhi |= (u32)(m >> 32) ^ (u32)sel;
            }
            plo = ~dlo; phi = ~dhi;
        }
        acc += __popc(plo & lt_lo) + __popc(phi & lt_hi);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const u32 n = 200000000u / 4096 * 4096;
    u64 *a, *b; u32* o;
    CK(hipMalloc(&a, (size_t)n * 8)); CK(hipMalloc(&b, (size_t)n * 8)); CK(hipMalloc(&o, 1 << 24));
    CK(hipMemset(a, 1, (size_t)n * 8)); CK(hipMemset(b, 0, (size_t)n * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const u32 tiles = n / 4096;
    auto report = [&](const char* name, float ms, double bytes) { printf("%-28s %8.3f ms  %8.1f GB/s\n", name, ms, bytes / ms * 1e-6); };
    float ms;
#define TIME(name, launch, bytes) do { for (int w = 0; w < 2; ++w) { launch; } CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) { launch; } CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); report(name, ms / 5, bytes); } while (0)
    TIME("copy u64", (k_copy<<<tiles, 256>>>(a, b, n)), (double)n * 16);
    TIME("segscatter  16 B runs", (k_segscatter<2, false><<<tiles, 256>>>(a, b, n, tiles)), (double)n * 16);
    TIME("segscatter  16 B runs xcd", (k_segscatter<2, true><<<tiles, 256>>>(a, b, n, tiles)), (double)n * 16);
    TIME("segscatter  32 B runs", (k_segscatter<4, false><<<tiles, 256>>>(a, b, n, tiles)), (double)n * 16);
    TIME("segscatter  32 B runs xcd", (k_segscatter<4, true><<<tiles, 256>>>(a, b, n, tiles)), (double)n * 16);
    TIME("segscatter  64 B runs", (k_segscatter<8, false><<<tiles, 256>>>(a, b, n, tiles)), (double)n * 16);
    TIME("segscatter  64 B runs xcd", (k_segscatter<8, true><<<tiles, 256>>>(a, b, n, tiles)), (double)n * 16);
    TIME("segscatter 128 B runs", (k_segscatter<16, false><<<tiles, 256>>>(a, b, n, tiles)), (double)n * 16);
    TIME("segscatter 128 B runs xcd", (k_segscatter<16, true><<<tiles, 256>>>(a, b, n, tiles)), (double)n * 16);
    {   // random keys for the staged replica
        u64* h = (u64*)malloc((size_t)n * 8);
        u64 x = 88172645463325252ull;
        for (u32 i = 0; i < n; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = x; }
        CK(hipMemcpy(a, h, (size_t)n * 8, hipMemcpyHostToDevice)); free(h);
    }
    TIME("stage0 load+exchange+store", (k_stage<0><<<tiles, 256>>>(a, b, n, 8)), (double)n * 16);
    TIME("stage1 + match arithmetic", (k_stage<1><<<tiles, 256>>>(a, b, n, 8)), (double)n * 16);
    TIME("stage2 + LDS atomics/shfl", (k_stage<2><<<tiles, 256>>>(a, b, n, 8)), (double)n * 16);
    TIME("stage3 full ranking", (k_stage<3><<<tiles, 256>>>(a, b, n, 8)), (double)n * 16);
    const int iters = 2048, blocks = 4096;
    for (int mode = 0; mode < 2; ++mode) {
        for (int w = 0; w < 2; ++w) { if (mode == 0) k_match<0><<<blocks, 256>>>(o, iters); else k_match<1><<<blocks, 256>>>(o, iters); }
        CK(hipEventRecord(e0));
        for (int r = 0; r < 3; ++r) { if (mode == 0) k_match<0><<<blocks, 256>>>(o, iters); else k_match<1><<<blocks, 256>>>(o, iters); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double wave_matches = (double)blocks * 4 * iters * 3;
        // 1024 SIMDs; ns per match per SIMD
        printf("match %c: %.3f ms per launch, %.2f ns per wave-match per SIMD (%.1f cycles at 2.4 GHz)\n", mode ? 'B' : 'A', ms / 3,
               ms * 1e6 / (wave_matches / 1024), ms * 1e6 / (wave_matches / 1024) * 2.4);
    }
    return 0;
}
