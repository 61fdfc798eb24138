// tools/valu_microbench.hip -- issue-rate probe for the VALU instruction forms the PairHMM
// kernel is built from (development aid; not part of libmgx.so).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_microbench.hip -o /tmp/valu_microbench && /tmp/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

constexpr int kIters = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 0.999f, c = 1e-7f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    int i0 = threadIdx.x, i1 = 3;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {  // v_fma_f32, 8 independent chains
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if constexpr (MODE == 1) {  // v_pk_fma_f32, 4 independent chains (8 floats)
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        } else if constexpr (MODE == 2) {  // v_mul_f32_dpp row_shr:1 (shift fused into the multiply)
            REP8(asm volatile("v_mul_f32_dpp %0, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mul_f32_dpp %2, %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mul_f32_dpp %4, %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %5, %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mul_f32_dpp %6, %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %7, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if constexpr (MODE == 3) {  // v_mov_b32_dpp wave_shr:1
            REP8(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if constexpr (MODE == 4) {  // v_cmp_eq_u32 -> sgpr pair + v_cndmask (2 instr per item)
            REP8(asm volatile("v_cmp_eq_u32 s[10:11], %8, %9\n v_cndmask_b32 %0, %0, %1, s[10:11]\n v_cmp_eq_u32 s[12:13], %8, %9\n v_cndmask_b32 %2, %2, %3, s[12:13]\n"
                              "v_cmp_eq_u32 s[10:11], %8, %9\n v_cndmask_b32 %4, %4, %5, s[10:11]\n v_cmp_eq_u32 s[12:13], %8, %9\n v_cndmask_b32 %6, %6, %7, s[12:13]\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0), "v"(i1) : "s10", "s11", "s12", "s13");)
        } else if constexpr (MODE == 5) {  // v_mul_f32 plain
            REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if constexpr (MODE == 6) {  // v_mov_b32_dpp row_shr:1
            REP8(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if constexpr (MODE == 7) {  // v_fma_f64, 4 chains
            double d0 = a0, d1 = a1, d2 = a2, d3 = a3, db = 0.999, dc = 1e-9;
            REP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                              "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db), "v"(dc));)
            a0 += (float)(d0 + d1 + d2 + d3);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE>
void run(const char* name, int waves_per_simd, double lanes_per_instr) {
    int dev_cus = 256;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); dev_cus = prop.multiProcessorCount;
    int blocks = dev_cus * waves_per_simd;   // 256 threads = 4 waves = 1 wave per SIMD per block
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, kIters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_wave = (double)kIters * 64;             // 8 x 8 per iteration
    double wave_instr = instr_per_wave * blocks * 4;
    double per_simd_per_s = wave_instr / (dev_cus * 4.0) / (ms * 1e-3);
    printf("%-34s waves/SIMD=%d  %.3f ms  %.2f G wave-instr/s/SIMD  => %.2f cycles/instr @2.4GHz, %.1f T lane-ops/s\n",
           name, waves_per_simd, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s, wave_instr * lanes_per_instr / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w, 64);
        run<1>("v_pk_fma_f32 (2 floats/lane)", w, 128);
        run<5>("v_mul_f32", w, 64);
        run<2>("v_mul_f32_dpp row_shr:1", w, 64);
        run<6>("v_mov_b32_dpp row_shr:1", w, 64);
        run<3>("v_mov_b32_dpp wave_shr:1", w, 64);
        run<4>("v_cmp_eq_u32 + v_cndmask_b32", w, 64);
        run<7>("v_fma_f64", w, 64);
    }
    return 0;
}
