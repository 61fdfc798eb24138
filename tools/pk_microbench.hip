// tools/pk_microbench.hip -- does packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) buy issue rate on gfx950?
// (development aid; not part of libmgx.so).  VERDICT r2 item 3 asks for a hand-packed PairHMM column; this probe prices
// the instruction mix of one column step of pairhmm_fwd<float,16,8> (8 rows per lane, the 7-operation cell of
// DESIGN.md 3.2) in its scalar form and in a packed form (rows k and k+4 of a lane in one aligned register pair),
// with the real shader clock measured in the kernel (s_memtime ticks per s_memrealtime tick of 100 MHz).
//   hipcc -O3 --offload-arch=gfx950 tools/pk_microbench.hip -o tools/bin/pk_microbench && tools/bin/pk_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

constexpr int kIters = 2048;

// MODE 0: 64 x v_fma_f32              1: 32 x v_pk_fma_f32       2: 32 x v_pk_mul_f32      3: 32 x v_pk_add_f32
// MODE 4: scalar column step (8 rows: 8 x [fma add mul | mul fma | mul fma] + 3 DPP + 1 ds_read_b64 x 4)
// MODE 5: packed column step (4 row pairs: 4 x [pk_fma pk_add pk_mul | pk_mul pk_fma | pk_mul pk_fma] + 3 DPP + same reads)
// MODE 6: packed M and Y, scalar X chain (what VERDICT r2 item 3 describes)
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, uint64_t* clk, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0.999f + i * 1e-7f;
    __syncthreads();
    const int t = threadIdx.x;
    float m[8], x[8], y[8], pmm[8], a[8], b[8], c[8], d[8], e[8];
    for (int s = 0; s < 8; ++s) {
        m[s] = 1.0f + t * 1e-3f + s; x[s] = 0.5f + s; y[s] = 0.25f + s;
        pmm[s] = 0.99f - s * 1e-3f; a[s] = 1e-3f + s * 1e-5f; b[s] = 0.1f; c[s] = 2e-3f; d[s] = 0.1f; e[s] = 0.999f;
    }
    f2 M[4], X[4], Y[4], PMM[4], A[4], B[4], C[4], D[4], E[4];
    for (int k = 0; k < 4; ++k) {
        M[k] = f2{m[k], m[k + 4]}; X[k] = f2{x[k], x[k + 4]}; Y[k] = f2{y[k], y[k + 4]};
        PMM[k] = f2{pmm[k], pmm[k + 4]}; A[k] = f2{a[k], a[k + 4]}; B[k] = f2{b[k], b[k + 4]};
        C[k] = f2{c[k], c[k + 4]}; D[k] = f2{d[k], d[k + 4]}; E[k] = f2{e[k], e[k + 4]};
    }
    const uint32_t lds_addr = (uint32_t)(t & 63) * 8;
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(m[0]), "+v"(m[1]), "+v"(m[2]), "+v"(m[3]), "+v"(m[4]), "+v"(m[5]), "+v"(m[6]), "+v"(m[7]) : "v"(b[0]), "v"(a[0]));)
        } else if constexpr (MODE == 1) {
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              : "+v"(M[0]), "+v"(M[1]), "+v"(M[2]), "+v"(M[3]) : "v"(B[0]), "v"(A[0]));)
        } else if constexpr (MODE == 2) {
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                              : "+v"(M[0]), "+v"(M[1]), "+v"(M[2]), "+v"(M[3]) : "v"(E[0]));)
        } else if constexpr (MODE == 3) {
            REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                              : "+v"(M[0]), "+v"(M[1]), "+v"(M[2]), "+v"(M[3]) : "v"(A[0]));)
        } else if constexpr (MODE == 4) {
            // one column: emissions of 8 rows (4 x ds_read_b64), 3 DPP hand-offs, 8 x 7 scalar operations (compiler-scheduled)
            f2 em[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) em[k] = *(const f2*)((const char*)lds + lds_addr + k * 512 + (it & 1) * 2048);
            float mu = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m[7]), 0x111, 0xf, 0xf, false));
            float xu = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x[7]), 0x111, 0xf, 0xf, false));
            float yu = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, y[7]), 0x111, 0xf, 0xf, false));
            float mn[8], xn[8], yn[8];
#pragma unroll
            for (int s = 7; s >= 0; --s) {
                float m2 = s ? m[s - 1] : mu, x2 = s ? x[s - 1] : xu, y2 = s ? y[s - 1] : yu;
                mn[s] = (__builtin_fmaf(pmm[s], m2, x2) + y2) * ((s & 1) ? em[s >> 1].y : em[s >> 1].x);
                yn[s] = __builtin_fmaf(d[s], y[s], c[s] * m[s]);
            }
            float xp = xu, mp = mu;
#pragma unroll
            for (int s = 0; s < 8; ++s) {                      // the serial chain down the rows of one column
                xn[s] = __builtin_fmaf(b[s], xp, a[s] * mp);
                xp = xn[s]; mp = mn[s];
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) { m[s] = mn[s]; x[s] = xn[s]; y[s] = yn[s]; }
        } else {
            f2 em[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) em[k] = *(const f2*)((const char*)lds + lds_addr + k * 512 + (it & 1) * 2048);
            float mu = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, M[3].y), 0x111, 0xf, 0xf, false));
            float xu = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, X[3].y), 0x111, 0xf, 0xf, false));
            float yu = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, Y[3].y), 0x111, 0xf, 0xf, false));
            f2 Mn[4], Xn[4], Yn[4];
            // rows k and k+4 in one pair; "row above" of pair k is pair k-1 (pair 0: the DPP value and the skewed half)
            f2 Mu = f2{mu, M[3].x}, Xu = f2{xu, X[3].x}, Yu = f2{yu, Y[3].x};
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                f2 M2 = k ? M[k - 1] : Mu, X2 = k ? X[k - 1] : Xu, Y2 = k ? Y[k - 1] : Yu;
                Mn[k] = (__builtin_elementwise_fma(PMM[k], M2, X2) + Y2) * em[k];
                Yn[k] = __builtin_elementwise_fma(D[k], Y[k], C[k] * M[k]);
            }
            if constexpr (MODE == 5) {
                f2 Xp = Xu, Mp = Mu;
#pragma unroll
                for (int k = 0; k < 4; ++k) {                  // two interleaved chains of four (column-skewed halves)
                    Xn[k] = __builtin_elementwise_fma(B[k], Xp, A[k] * Mp);
                    Xp = Xn[k]; Mp = Mn[k];
                }
            } else {
                float xp = xu, mp = mu;
#pragma unroll
                for (int s = 0; s < 8; ++s) {                  // scalar chain over the halves of the packed registers
                    float av = (s < 4) ? A[s].x : A[s - 4].y, bv = (s < 4) ? B[s].x : B[s - 4].y;
                    float r = __builtin_fmaf(bv, xp, av * mp);
                    if (s < 4) Xn[s].x = r; else Xn[s - 4].y = r;
                    xp = r; mp = (s < 4) ? Mn[s].x : Mn[s - 4].y;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { M[k] = Mn[k]; X[k] = Xn[k]; Y[k] = Yn[k]; }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0;
    for (int s = 0; s < 8; ++s) acc += m[s] + x[s] + y[s];
    for (int k = 0; k < 4; ++k) acc += M[k].x + M[k].y + X[k].x + X[k].y + Y[k].x + Y[k].y;
    out[blockIdx.x * blockDim.x + t] = acc;
    if (t == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int waves_per_simd, double instr_per_iter, double cells_per_iter) {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    int blocks = cus * waves_per_simd;
    float* out; uint64_t* clk;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&clk, (size_t)blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) probe<MODE><<<blocks, 256>>>(out, clk, kIters);      // let the clock settle
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, clk, kIters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(2 * blocks);
    hipMemcpy(h.data(), clk, (size_t)blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> ghz(blocks), cyc(blocks);
    for (int i = 0; i < blocks; ++i) { ghz[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; cyc[i] = (double)h[2 * i]; }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    double clock = ghz[blocks / 2], cycles = cyc[blocks / 2];
    // a SIMD runs `waves_per_simd` waves; cycles per wave-instruction as the SIMD sees it
    double per_instr = cycles / (kIters * instr_per_iter * waves_per_simd);
    printf("%-44s waves/SIMD=%d  %.3f ms  clock %.2f GHz  %.2f SIMD-cycles/instr", name, waves_per_simd, ms, clock, per_instr);
    if (cells_per_iter > 0) printf("  %.2f SIMD-cycles per 8-row column (64 lanes)", cycles / (kIters * waves_per_simd));
    printf("\n");
    hipFree(out); hipFree(clk);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32 x64", w, 64, 0);
        run<1>("v_pk_fma_f32 x32", w, 32, 0);
        run<2>("v_pk_mul_f32 x32", w, 32, 0);
        run<3>("v_pk_add_f32 x32", w, 32, 0);
        run<4>("column step, scalar (56 VALU + 11 mov + 3 DPP; 2 LDS)", w, 70, 8);
        run<5>("column step, packed M/X/Y (28 pk + 8 mov + 3 DPP; 2 LDS)", w, 39, 8);
        run<6>("column step, packed M/Y, scalar X (20 pk + 14 + 7 mov + 3 DPP)", w, 44, 8);
    }
    return 0;
}
