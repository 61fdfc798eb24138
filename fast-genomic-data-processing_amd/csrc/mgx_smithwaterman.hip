// mgx_smithwaterman.hip -- Smith-Waterman with back-trace on gfx950 (C ABI: include/mgx_smithwaterman.h).
//
// What the reference does per pair with AVX anti-diagonals (intel/smithwaterman/PairWiseSW.h) is done
// here with one wavefront per pair and the machinery of the PairHMM kernel: lane l owns RPL
// consecutive rows of the matrix and walks the columns, one anti-diagonal of lanes per step; the
// bottom row of a lane (H and F of the current column) reaches the next lane through a lane shift.
// Integer DP, so results are the reference's exactly:
//   * cell recurrence and tie rules: PairWiseSW.h:31-66 (E/F prefer extension on ties; H prefers the
//     diagonal, then E, then F; back-trace byte = op | INSERT_EXT | DELETE_EXT)
//   * boundaries per overhang strategy: PairWiseSW.h:243-253
//   * best end cell among equal scores, replayed in anti-diagonal order: PairWiseSW.h:256-285
//   * back-trace state machine, leading / trailing overhang, merge: PairWiseSW.h:299-408
//   * CIGAR text and its capacity rule (host, next to the caller's buffer): PairWiseSW.h:410-444
//
// Back-trace bytes are laid out by (step, lane, row-in-lane) so that one step of a wavefront stores
// 64 * RPL consecutive bytes; the trace kernels convert (i, j) back to that index (BtView).
//
// Two fill kernels (round 3):
//   k_sw_fill16   packed 16-bit scores, two pairs per lane group, the four decisions of a cell taken from the sign bits of four
//                 packed differences, back-trace nibbles, all row classes in one launch -- for every pair whose scores provably fit
//                 16 bits (I16Rule; the realignment workload of Mutect2 always does): 12.6 instructions per cell
//   k_sw_fill     32-bit scores, one pair per lane group, compare + select per decision, back-trace bytes, one launch per row class
//                 -- everything else (long references with long alternates, large scoring parameters); optionally with the lanes
//                 over the alternate sequence (TR; measured slower, opt-in)
// Both write what the trace kernels read through BtView / LastRow / LastCol; a batch may mix them pair by pair.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <memory>
#include <new>
#include <thread>
#include <vector>

#include "../../include/mgx_smithwaterman.h"
#include "mgx_common.h"

using mgx::set_error;

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) { set_error("%s: %s", #expr, hipGetErrorString(e_)); return -EIO; } \
    } while (0)

namespace {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

constexpr int kOpMatch = 0, kOpInsert = 1, kOpDelete = 2, kInsertExt = 4, kDeleteExt = 8;
constexpr int kMinCutoff = -100000000;           // MATRIX_MIN_CUTOFF, smithwaterman_common.h:73
constexpr int kLowInit = INT32_MIN / 2;          // LOW_INIT_VALUE, smithwaterman_common.h:74
constexpr int kMaxRef = 2048;                    // 64 lanes x 32 rows

struct SwJob {
    u64 off1, off2;      // into the uploaded ref / alt bytes
    u64 bt_off;          // back-trace arena, bytes
    u64 sc_off;          // score arena (int32): last_row[ncol + 1] then last_col[nrow + 1]
    u64 el_off;          // element arena (int16): 2 * (len1 + len2 + 2)
    u32 len1, len2, rpl, strategy, out_index;
    u32 g;               // bits 0-7: lanes per pair (32 or 64) | kJobTr | kJobI16 | kJobHalf
    u64 lr_off;          // 16-bit fill: the last lane's H values per swept position (back-trace arena, bytes)
    u64 lc_off;          // 16-bit fill: every lane's packed H values at the pair's last swept position
    int32_t low16;       // 16-bit fill: this pair's stand-in for LOW_INIT_VALUE
    u32 pad_;
};
constexpr u32 kJobTr = 0x100u;      // k_sw_fill<.., TR>: the lanes own the alternate sequence
constexpr u32 kJobI16 = 0x200u;     // k_sw_fill16: packed 16-bit scores, two pairs per lane group, back-trace nibbles
constexpr u32 kJobHalf = 0x400u;    // k_sw_fill16: this pair is the high half of its lane group
constexpr u32 kNoOutput = 0xFFFFFFFFu;   // out_index of a filler job (a 16-bit class is padded to whole wavefronts)

struct SwResult { int32_t score, max_i, max_j, offset, n_elems; };
// back-trace bytes of one lane and step: RPL of them, in a slot of whole 32-bit words from three positions per lane up (a 5-, 6- or
// 7-byte slot would be stored byte by byte)
__host__ __device__ constexpr int bt_slot(int rpl) { return rpl <= 2 ? rpl : (rpl + 3) & ~3; }
struct SwParams { int match, mismatch, open, extend; };

// G lanes per pair (two pairs share a wavefront when G = 32), RPL rows per lane.  Back-trace byte:
// bits 0-1 op, bit 2 INSERT_EXT, bit 3 DELETE_EXT (PairWiseSW.h:31-66).  (Storing the length of the
// diagonal match run in the spare bits, so that the trace crosses a run in one step, was measured:
// trace 0.58 -> 0.38 ms, fill 1.14 -> 1.42 ms per 20 000 pairs -- not kept.)
// TR (round 3): the lanes own the ALTERNATE sequence's positions (columns j) and sweep the reference (rows i) -- the
// transposed recurrence.  The reads realigned after PairHMM are 100-151 bases against haplotype windows of 250-400: with the
// lanes over the reference, 64 lanes x 8 rows hold 250-400 rows (49-78 % of the row slots) and a pair takes ncol + 63 steps of
// which ncol do work (66 %); with the lanes over the read, 32 lanes x 4-5 positions hold 100-151 (78-100 %) and a pair takes
// nrow + 31 steps of which nrow do work (91 %), two pairs per wavefront.  Every cell computes the same values with the same tie
// rules: E (horizontal gap) is the state carried along the sweep when the lanes own rows and the chain handed from position to
// position when they own columns, F (vertical gap) the other way round; H still prefers the diagonal, then E, then F.
template <int G, int RPL, bool TR>
__global__ __launch_bounds__(64) void k_sw_fill(const SwJob* __restrict__ jobs, u32 n_jobs, const u8* __restrict__ s1, const u8* __restrict__ s2,
                                                u8* __restrict__ bt, int32_t* __restrict__ sc, u32 lds_stride, SwParams P) {
    extern __shared__ u8 sh_all[];                    // the swept sequences: one global load per step would
    constexpr int GPW = 64 / G;                       // put ~200 serial HBM latencies on every wavefront's path
    const int grp = threadIdx.x / G, lane = threadIdx.x % G;
    const u32 job_idx = blockIdx.x * GPW + grp;
    const bool live = job_idx < n_jobs;
    SwJob J{};
    if (live) J = jobs[job_idx];
    const int nrow = (int)J.len1, ncol = (int)J.len2;
    const int n_own = TR ? ncol : nrow, n_swp = TR ? nrow : ncol;      // positions spread over the lanes | positions swept
    const bool indel = J.strategy == MGX_SW_INDEL || J.strategy == MGX_SW_LEADING_INDEL;
    const u8* own = TR ? s2 + J.off2 : s1 + J.off1;
    const u8* swp = TR ? s1 + J.off1 : s2 + J.off2;
    int32_t* last_row = sc + J.sc_off;               // H(nrow, j), j = 1 .. ncol
    int32_t* last_col = last_row + ncol + 1;         // H(i, ncol), i = 1 .. nrow
    int32_t* const edge_own = TR ? last_col : last_row;      // written step by step by the lane that owns the last position
    int32_t* const edge_swp = TR ? last_row : last_col;      // the registers after the last step
    u8* btp = bt + J.bt_off;
    u8* sh_b = sh_all + (size_t)grp * lds_stride;
    for (int x = lane; x < n_swp; x += G) sh_b[x] = swp[x];
    __syncthreads();
    const int r0 = lane * RPL;                       // this lane owns positions r0+1 .. r0+RPL (1-based)
    int sa[RPL], hl[RPL], gs[RPL];                   // gs: the gap state along the sweep (E when the lanes own rows, F when columns)
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        const int i = r0 + k + 1;
        sa[k] = (live && i <= n_own) ? (int)own[i - 1] : 256;     // padding positions never match and feed nothing beyond them
        hl[k] = indel ? P.open + (i - 1) * P.extend : 0;          // H(i, 0) / H(0, j), PairWiseSW.h:243-253
        gs[k] = kLowInit;                                         // E(i, 0) / F(0, j)
    }
    int diag_in = r0 == 0 ? 0 : (indel ? P.open + (r0 - 1) * P.extend : 0);   // H(r0, 0) / H(0, r0)
    const int n_lanes = live ? (n_own + RPL - 1) / RPL : 0;
    const int steps = live ? n_swp + n_lanes - 1 : 0;
    int steps_w = steps;                              // the wavefront walks to its longest pair
    if constexpr (GPW > 1) {
        const int other = __shfl_xor(steps, G, 64);
        steps_w = other > steps ? other : steps;
    }
    const int lane_last = live ? (n_own - 1) / RPL : -1, k_last = live ? (n_own - 1) % RPL : 0;
    int send_h = 0, send_g = kLowInit;
    for (int t = 1; t <= steps_w; ++t) {
        // what the lane before computed for this sweep position one step ago: H and the chained gap at its last position
        const int rh = __shfl_up(send_h, 1, G), rg = __shfl_up(send_g, 1, G);
        const int j = t - lane;                       // the sweep position of this lane at this step
        if (j >= 1 && j <= n_swp && lane < n_lanes) {
            int up_h, up_g, diag;
            if (lane == 0) {
                up_h = indel ? P.open + (j - 1) * P.extend : 0;                           // H(0, j) / H(i, 0)
                up_g = kLowInit;                                                          // F(0, j) / E(i, 0)
                diag = j == 1 ? 0 : (indel ? P.open + (j - 2) * P.extend : 0);            // the boundary one position back, H(0,0) = 0
            } else { up_h = rh; up_g = rg; diag = diag_in; }
            diag_in = up_h;                                          // this step's neighbour value is the next step's diagonal
            const int c2 = (int)sh_b[j - 1];
            u32 packed[(RPL + 3) / 4];
#pragma unroll
            for (int q = 0; q < (RPL + 3) / 4; ++q) packed[q] = 0;
#pragma unroll
            for (int k = 0; k < RPL; ++k) {
                const int open_s = hl[k] + P.open, ext_s = gs[k] + P.extend;         // along the sweep
                const int gs_new = max(open_s, ext_s);
                const int open_c = up_h + P.open, ext_c = up_g + P.extend;           // along the lanes' own positions
                const int gc_new = max(ext_c, open_c);
                // E is the horizontal gap (INSERT), F the vertical one (DELETE); ties prefer the extension
                int ext = 0;
                if (!(open_s > ext_s)) ext |= TR ? kDeleteExt : kInsertExt;
                if (!(open_c > ext_c)) ext |= TR ? kInsertExt : kDeleteExt;
                const int ee = TR ? gc_new : gs_new, ff = TR ? gs_new : gc_new;
                int h = max(diag + (sa[k] == c2 ? P.match : P.mismatch), kMinCutoff);
                int op = kOpMatch;
                if (ee > h) { op = kOpInsert; h = ee; }
                if (ff > h) { op = kOpDelete; h = ff; }
                diag = hl[k]; hl[k] = h; gs[k] = gs_new; up_h = h; up_g = gc_new;
                packed[k >> 2] |= (u32)(op | ext) << ((k & 3) * 8);
            }
            send_h = up_h; send_g = up_g;
            if (lane == lane_last) {
                int v = hl[0];
#pragma unroll
                for (int k = 1; k < RPL; ++k) if (k == k_last) v = hl[k];
                edge_own[j] = v;
            }
            constexpr int SLOT = bt_slot(RPL);
            u8* dst = btp + ((size_t)t * G + lane) * SLOT;
            if constexpr (RPL == 1) dst[0] = (u8)packed[0];
            else if constexpr (RPL == 2) *reinterpret_cast<uint16_t*>(dst) = (uint16_t)packed[0];
            else {
#pragma unroll
                for (int q = 0; q < SLOT / 4; ++q) reinterpret_cast<u32*>(dst)[q] = packed[q];
            }
        }
    }
    // a lane's last active step is the last sweep position: its registers now hold H there (kept out of the loop --
    // a guarded store per cell and step doubled the instruction count of the sweep)
    if (lane < n_lanes) {
#pragma unroll
        for (int k = 0; k < RPL; ++k) if (r0 + k + 1 <= n_own) edge_swp[r0 + k + 1] = hl[k];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same fill with packed 16-bit scores (round 3).  The 32-bit sweep issues ~26 integer instructions per cell (compare + select for
// every tie rule and flag); here a lane group carries TWO pairs, one in each half of every register (v_pk_add / v_pk_max / v_pk_sub _i16),
// and the four decisions of a cell -- E opened, F opened, E beats the diagonal, F beats both -- are the SIGN BITS of four packed
// differences, merged with v_bfi into one nibble per pair: ~25 instructions per two cells.  Values are the reference's exactly as long
// as nothing wraps: the host admits a pair only when every real value and every difference of two of them fits 16 bits
// (I16Rule below: bounds from the lengths and the scoring parameters; LOW_INIT_VALUE becomes a per-pair value below every real one),
// everything else takes k_sw_fill.  Cells outside a pair's own matrix (the other half is longer, padding rows) may wrap: nothing real
// reads them.
//   * back-trace: one 32-bit word per lane, step and four rows: low half = the first pair's four nibbles (row 4q in bits 0-3),
//     high half = the second pair's; nibble = {8: E opened, 4: F opened, 2: E > diagonal, 1: F > max(diagonal, E)}
//   * H(nrow, j): the lane that owns a pair's last row stores its RPL packed H values per step (three or four store instructions;
//     picking the one row out of the registers would cost RPL selects per step); the trace kernels read row (nrow - 1) % RPL of it
//   * H(i, ncol): every lane stores its RPL packed H values when it passes the pair's last column (the other pair may sweep on)
//   * every class of a batch runs in ONE launch (a wave-uniform switch over RPL; classes are padded to whole wavefronts with filler
//     jobs): 20 000 pairs are 5 000 wavefronts -- per-class launches of a few hundred wavefronts each would leave the device idle.
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 pk2(int lo, int hi) { s16x2 v; v.x = (short)lo; v.y = (short)hi; return v; }
__device__ __forceinline__ u32 bits(s16x2 v) { return __builtin_bit_cast(u32, v); }
__device__ __forceinline__ s16x2 s16(u32 v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ s16x2 pk_max(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
// (a & mask) | (b & ~mask) in one instruction; written out, the compiler splits it into v_and + v_and_or (no 32-bit literals in VOP3 on gfx9)
__device__ __forceinline__ u32 bfi(u32 mask, u32 a, u32 b) { u32 r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(a), "v"(b)); return r; }
// 1 in every half whose bases differ (as a compare it becomes two 16-bit compares, two selects and a v_perm)
__device__ __forceinline__ u32 pk_min_u16(u32 a, u32 b) { u32 r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ s16x2 pk_mad(u32 a, s16x2 b, s16x2 c) { u32 r; asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(bits(b)), "v"(bits(c))); return s16(r); }
__host__ __device__ constexpr int bt_slot16(int rpl) { return 4 * ((rpl + 3) / 4); }

template <int G, int RPL>
__device__ __forceinline__ void sw_fill16_body(const SwJob& JA, const SwJob& JB, const int lane, const u8* __restrict__ s1, const u8* __restrict__ s2,
                                               u8* __restrict__ bt, int32_t* __restrict__ sc, u32* sh, const SwParams P) {
    constexpr int GPW = 64 / G;
    constexpr int W = (RPL + 3) / 4;                  // back-trace words per lane and step
    const int nrowA = (int)JA.len1, ncolA = (int)JA.len2, nrowB = (int)JB.len1, ncolB = (int)JB.len2;
    const int n_swp = max(ncolA, ncolB);
    const bool indelA = JA.strategy == MGX_SW_INDEL || JA.strategy == MGX_SW_LEADING_INDEL;
    const bool indelB = JB.strategy == MGX_SW_INDEL || JB.strategy == MGX_SW_LEADING_INDEL;
    const u8* rowsA = s1 + JA.off1; const u8* rowsB = s1 + JB.off1;
    const u8* colsA = s2 + JA.off2; const u8* colsB = s2 + JB.off2;
    for (int x = lane; x < n_swp; x += G) sh[x] = (x < ncolA ? (u32)colsA[x] : 0u) | ((x < ncolB ? (u32)colsB[x] : 0u) << 16);
    __syncthreads();
    const int r0 = lane * RPL;
    const s16x2 vopen = pk2(P.open, P.open), vext = pk2(P.extend, P.extend), vmatch = pk2(P.match, P.match);
    const s16x2 vdelta = pk2(P.mismatch - P.match, P.mismatch - P.match);
    const s16x2 vlow = pk2(JA.low16, JB.low16);
    u32 sa[RPL];
    s16x2 hl[RPL], gs[RPL];
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        const int i = r0 + k + 1;
        sa[k] = (i <= nrowA ? (u32)rowsA[i - 1] : 256u) | ((i <= nrowB ? (u32)rowsB[i - 1] : 256u) << 16);
        hl[k] = pk2(indelA ? P.open + (i - 1) * P.extend : 0, indelB ? P.open + (i - 1) * P.extend : 0);     // H(i, 0)
        gs[k] = vlow;                                                                                       // E(i, 0)
    }
    s16x2 diag_in = r0 == 0 ? pk2(0, 0) : pk2(indelA ? P.open + (r0 - 1) * P.extend : 0, indelB ? P.open + (r0 - 1) * P.extend : 0);   // H(r0, 0)
    s16x2 bnd = pk2(indelA ? P.open : 0, indelB ? P.open : 0);                      // lane 0: H(0, j), starting at j = 1
    const s16x2 bstep = pk2(indelA ? P.extend : 0, indelB ? P.extend : 0);
    const int n_lanes = max((nrowA + RPL - 1) / RPL, (nrowB + RPL - 1) / RPL);
    const int steps = n_lanes > 0 ? n_swp + n_lanes - 1 : 0;
    int steps_w = steps;
    if constexpr (GPW > 1) {
        const int other = __shfl_xor(steps, G, 64);
        steps_w = other > steps ? other : steps;
    }
    const int lane_lastA = nrowA > 0 ? (nrowA - 1) / RPL : -1, lane_lastB = nrowB > 0 ? (nrowB - 1) / RPL : -1;
    u32* const lrA = reinterpret_cast<u32*>(bt + JA.lr_off);
    u32* const lrB = reinterpret_cast<u32*>(bt + JB.lr_off);
    const bool lr_apart = JA.lr_off != JB.lr_off;
    u32* const lcA = reinterpret_cast<u32*>(bt + JA.lc_off);
    u32* const lcB = reinterpret_cast<u32*>(bt + JB.lc_off);
    const bool lc_apart = JA.lc_off != JB.lc_off;
    u8* const btp = bt + JA.bt_off;
    s16x2 send_h = pk2(0, 0), send_g = vlow;
    for (int t = 1; t <= steps_w; ++t) {
        const u32 rh = (u32)__builtin_amdgcn_update_dpp(0, (int)bits(send_h), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        const u32 rg = (u32)__builtin_amdgcn_update_dpp(0, (int)bits(send_g), 0x138, 0xf, 0xf, false);
        const int j = t - lane;
        if (j >= 1 && j <= n_swp && lane < n_lanes) {
            s16x2 up_h = lane == 0 ? bnd : s16(rh);                  // H(r0, j)
            s16x2 up_g = lane == 0 ? vlow : s16(rg);                 // F(r0, j)
            bnd += bstep;
            s16x2 diag = diag_in;                                    // H(r0, j - 1)
            diag_in = up_h;
            const u32 c2 = sh[j - 1];
            u32 word[W];
#pragma unroll
            for (int q = 0; q < W; ++q) word[q] = 0;
#pragma unroll
            for (int k = 0; k < RPL; ++k) {
                const s16x2 open_s = hl[k] + vopen, ext_s = gs[k] + vext;
                const s16x2 ee = pk_max(open_s, ext_s);
                const s16x2 open_c = up_h + vopen, ext_c = up_g + vext;
                const s16x2 ff = pk_max(ext_c, open_c);
                const u32 ne = pk_min_u16(sa[k] ^ c2, 0x00010001u);                                    // 1 where the bases differ
                const s16x2 h0 = diag + pk_mad(ne, vdelta, vmatch);
                const s16x2 m1 = pk_max(h0, ee);
                const s16x2 h = pk_max(m1, ff);
                // sign bit set: E opened (no INSERT_EXT) | F opened (no DELETE_EXT) | ee > h0 | ff > max(h0, ee)
                const u32 t1 = bits(ext_s - open_s), t2 = bits(ext_c - open_c), t3 = bits(h0 - ee), t4 = bits(m1 - ff);
                const u32 u12 = bfi(0x80008000u, t1, t2 >> 1), u34 = bfi(0x80008000u, t3, t4 >> 1);
                const u32 nib = bfi(0xC000C000u, u12, u34 >> 2);
                word[k >> 2] = bfi(0xF000F000u, nib, word[k >> 2] >> 4);
                diag = hl[k]; hl[k] = h; gs[k] = ee; up_h = h; up_g = ff;
            }
            if constexpr (RPL % 4 != 0) word[W - 1] >>= 4 * (4 - RPL % 4);
            send_h = up_h; send_g = up_g;
            u32* dst = reinterpret_cast<u32*>(btp + ((size_t)t * G + lane) * (4 * W));
#pragma unroll
            for (int q = 0; q < W; ++q) dst[q] = word[q];
            if (lane == lane_lastA) {
                u32* d = lrA + (size_t)j * RPL;
#pragma unroll
                for (int k = 0; k < RPL; ++k) d[k] = bits(hl[k]);
            }
            if (lr_apart && lane == lane_lastB) {
                u32* d = lrB + (size_t)j * RPL;
#pragma unroll
                for (int k = 0; k < RPL; ++k) d[k] = bits(hl[k]);
            }
            if (j == ncolA) {
                u32* d = lcA + r0;
#pragma unroll
                for (int k = 0; k < RPL; ++k) d[k] = bits(hl[k]);
            }
            if (lc_apart && j == ncolB) {
                u32* d = lcB + r0;
#pragma unroll
                for (int k = 0; k < RPL; ++k) d[k] = bits(hl[k]);
            }
        }
    }
}

// jobs: pairs 2g and 2g + 1 share lane group g; both of one class (rpl), a class ends on a wavefront boundary.  BIG: the classes of
// more than 16 rows per lane, a kernel of their own (178 registers; the common one keeps 100 and twice the wavefronts per SIMD)
template <int G, bool BIG>
__global__ __launch_bounds__(64) void k_sw_fill16(const SwJob* __restrict__ jobs, u32 n_jobs, const u8* __restrict__ s1, const u8* __restrict__ s2,
                                                  u8* __restrict__ bt, int32_t* __restrict__ sc, u32 lds_stride, SwParams P) {
    extern __shared__ u32 sh16_all[];
    constexpr int GPW = 64 / G;
    const int grp = threadIdx.x / G, lane = threadIdx.x % G;
    const u32 g = blockIdx.x * GPW + grp;
    SwJob JA{}, JB{};
    if (2 * g < n_jobs) { JA = jobs[2 * g]; JB = jobs[2 * g + 1]; }
    u32* sh = sh16_all + (size_t)grp * lds_stride;
    const int rpl = __builtin_amdgcn_readfirstlane((int)jobs[2 * (size_t)blockIdx.x * GPW].rpl);       // the wavefront's class
#define MGX_SW16_CASE(r) case r: sw_fill16_body<G, r>(JA, JB, lane, s1, s2, bt, sc, sh, P); break;
    if constexpr (BIG) {
        switch (rpl) {
            MGX_SW16_CASE(20) MGX_SW16_CASE(24) MGX_SW16_CASE(32)
            default: break;
        }
    } else {
        switch (rpl) {
            MGX_SW16_CASE(1) MGX_SW16_CASE(2) MGX_SW16_CASE(3) MGX_SW16_CASE(4) MGX_SW16_CASE(5) MGX_SW16_CASE(6) MGX_SW16_CASE(7) MGX_SW16_CASE(8)
            MGX_SW16_CASE(9) MGX_SW16_CASE(10) MGX_SW16_CASE(11) MGX_SW16_CASE(12) MGX_SW16_CASE(13) MGX_SW16_CASE(14) MGX_SW16_CASE(16)
            default: break;
        }
    }
#undef MGX_SW16_CASE
}

// what the trace kernels read of a pair's back-trace, whichever kernel filled it: the byte of the 32-bit form (op | INSERT_EXT | DELETE_EXT)
struct BtView {
    const u8* p; int rpl, g, slot; bool tr, i16; int half;
    __device__ BtView(const SwJob& J, const u8* bt) : p(bt + J.bt_off), rpl((int)J.rpl), g((int)(J.g & 0xFFu)), tr((J.g & kJobTr) != 0),
                                                     i16((J.g & kJobI16) != 0), half((J.g & kJobHalf) ? 1 : 0) {
        slot = i16 ? bt_slot16(rpl) : bt_slot(rpl);
    }
    __device__ int cell(int i, int j) const {
        const int o = tr ? j : i, w = tr ? i : j;                      // position among the lanes | swept position
        const int l = (o - 1) / rpl, k = (o - 1) - l * rpl;
        const size_t base = ((size_t)(w + l) * g + l) * slot;
        if (!i16) return p[base + k];
        const int by = p[base + (k >> 2) * 4 + half * 2 + ((k & 3) >> 1)];
        const int nib = (k & 1) ? by >> 4 : by & 15;
        return ((nib & 1) ? kOpDelete : (nib & 2) ? kOpInsert : kOpMatch) | ((nib & 8) ? 0 : kInsertExt) | ((nib & 4) ? 0 : kDeleteExt);
    }
};
// H(nrow, j), j = 1 .. ncol
struct LastRow {
    const int32_t* p32; const u32* p16; int rpl, k_last, half;
    __device__ LastRow(const SwJob& J, const u8* bt, const int32_t* sc) : p32(sc + J.sc_off), p16(reinterpret_cast<const u32*>(bt + J.lr_off)),
                                                                            rpl((int)J.rpl), k_last(((int)J.len1 - 1) % (int)J.rpl), half((J.g & kJobHalf) ? 16 : 0) {
        if (!(J.g & kJobI16)) p16 = nullptr;
    }
    __device__ int operator[](int j) const { return p16 ? (int)(int16_t)(p16[(size_t)j * rpl + k_last] >> half) : p32[j]; }
};
// H(i, ncol), i = 1 .. nrow
struct LastCol {
    const int32_t* p32; const u32* p16; int half;
    __device__ LastCol(const SwJob& J, const u8* bt, const int32_t* sc) : p32(sc + J.sc_off + J.len2 + 1), p16(reinterpret_cast<const u32*>(bt + J.lc_off)),
                                                                            half((J.g & kJobHalf) ? 16 : 0) {
        if (!(J.g & kJobI16)) p16 = nullptr;
    }
    __device__ int operator[](int i) const { return p16 ? (int)(int16_t)(p16[i - 1] >> half) : p32[i]; }
};

// One lane per pair: best end cell, back-trace, merged element list (in the order getCIGAR holds it).
__global__ __launch_bounds__(64) void k_sw_trace(const SwJob* __restrict__ jobs, u32 n, const u8* __restrict__ bt,
                                                 const int32_t* __restrict__ sc, int16_t* __restrict__ elems_all, SwResult* __restrict__ res,
                                                 int16_t* __restrict__ compact, int n_compact) {
    const u32 p = blockIdx.x * 64 + threadIdx.x;
    if (p >= n) return;
    const SwJob J = jobs[p];
    if (J.out_index == kNoOutput) return;
    const int nrow = (int)J.len1, ncol = (int)J.len2, strategy = (int)J.strategy;
    const LastRow last_row(J, bt, sc);
    const LastCol last_col(J, bt, sc);
    const BtView view(J, bt);
    int16_t* el = elems_all + J.el_off;
    // PairWiseSW.h:256-285, in anti-diagonal order
    int best = INT32_MIN, mi = 0, mj = 0;
    for (int d = 1; d <= nrow + ncol; ++d) {
        if (d >= nrow + 1 && (strategy == MGX_SW_SOFTCLIP || strategy == MGX_SW_IGNORE)) {
            const int j = d - nrow;
            const int s = last_row[j];
            if (best < s || (best == s && abs(nrow - j) < abs(mi - mj))) { best = s; mi = nrow; mj = j; }
        }
        if (d >= ncol + 1) {
            const int i = d - ncol;
            const int s = last_col[i];
            if (best < s || (best == s && (mj == ncol || abs(i - ncol) <= abs(mi - mj)))) { best = s; mi = i; mj = ncol; }
        }
    }
    // PairWiseSW.h:299-408
    int i, j, m = 0;
    if (strategy == MGX_SW_INDEL) { i = nrow; j = ncol; }
    else if (strategy == MGX_SW_LEADING_INDEL) { i = mi; j = ncol; }
    else { i = mi; j = mj; }
    if (j < ncol) { el[0] = MGX_SW_SOFTCLIP; el[1] = (int16_t)(ncol - j); m = 1; }
    int state = 0;
    while (i > 0 && j > 0) {
        const int btr = view.cell(i, j);
        if (state == kInsertExt) { --j; el[2 * m - 1]++; state = btr & kInsertExt; }
        else if (state == kDeleteExt) { --i; el[2 * m - 1]++; state = btr & kDeleteExt; }
        else {
            const int op = btr & 3;
            if (op == kOpMatch) { --i; --j; el[2 * m] = kOpMatch; el[2 * m + 1] = 1; state = 0; ++m; }
            else if (op == kOpInsert) { --j; el[2 * m] = kOpInsert; el[2 * m + 1] = 1; state = btr & kInsertExt; ++m; }
            else { --i; el[2 * m] = kOpDelete; el[2 * m + 1] = 1; state = btr & kDeleteExt; ++m; }
        }
    }
    int offset;
    if (strategy == MGX_SW_SOFTCLIP) {
        if (j > 0) { el[2 * m] = MGX_SW_SOFTCLIP; el[2 * m + 1] = (int16_t)j; ++m; }
        offset = i;
    } else if (strategy == MGX_SW_IGNORE) {
        if (j > 0) { el[2 * m] = el[2 * (m - 1)]; el[2 * m + 1] = (int16_t)j; ++m; }
        offset = (int16_t)(i - j);
    } else {
        if (i > 0) { el[2 * m] = kOpDelete; el[2 * m + 1] = (int16_t)i; ++m; }
        else if (j > 0) { el[2 * m] = kOpInsert; el[2 * m + 1] = (int16_t)j; ++m; }
        offset = 0;
    }
    int w = 0;
    int16_t prev = el[0];
    for (int q = 1; q < m; ++q) {
        const int16_t cur = el[2 * q];
        if (cur == prev) el[2 * w + 1] = (int16_t)(el[2 * w + 1] + el[2 * q + 1]);
        else { ++w; el[2 * w] = cur; el[2 * w + 1] = el[2 * q + 1]; prev = cur; }
    }
    SwResult r; r.score = best; r.max_i = mi; r.max_j = mj; r.offset = offset; r.n_elems = w + 1;
    res[J.out_index] = r;
    // alignments have a handful of elements: those travel back in a fixed-size record per pair
    int16_t* ce = compact + (size_t)J.out_index * 2 * n_compact;
    for (int q = 0; q <= w && q < n_compact; ++q) { ce[2 * q] = el[2 * q]; ce[2 * q + 1] = el[2 * q + 1]; }
}

// The same with one WAVEFRONT per pair.  A lane per pair walks ~275 dependent byte loads of ~2 us each with a third of
// a wavefront per SIMD to hide them behind; here the 64 lanes fetch the next 64 cells of the current DIAGONAL in one
// gather (lane t: cell (i - t, j - t)) and the walk consumes them from registers, so an alignment of a few hundred
// matches with a handful of gaps costs a handful of memory round trips.  Inside a gap (extension states) cells are
// loaded one at a time as before.  The walk itself is wave-uniform; elements are merged on the fly (run-length form,
// which is what the reference's final merge pass produces, PairWiseSW.h:390-408) and written by lane 0.
__global__ __launch_bounds__(64) void k_sw_trace_wave(const SwJob* __restrict__ jobs, u32 n, const u8* __restrict__ bt,
                                                      const int32_t* __restrict__ sc, int16_t* __restrict__ elems_all, SwResult* __restrict__ res,
                                                      int16_t* __restrict__ compact, int n_compact) {
    const u32 p = blockIdx.x;
    if (p >= n) return;
    const int lane = threadIdx.x;
    const SwJob J = jobs[p];
    if (J.out_index == kNoOutput) return;
    const int nrow = (int)J.len1, ncol = (int)J.len2, strategy = (int)J.strategy;
    const LastRow last_row(J, bt, sc);
    const LastCol last_col(J, bt, sc);
    const BtView view(J, bt);
    int16_t* el = elems_all + J.el_off;
    const bool use_row = strategy == MGX_SW_SOFTCLIP || strategy == MGX_SW_IGNORE;
    // ---- best end cell (PairWiseSW.h:256-285).  The maximum first, in parallel; then the cells that reach it are
    //      replayed in anti-diagonal order (row candidate before column candidate), which is all the tie rules see.
    int best = INT32_MIN;
    int mi = 0, mj = 0;
    bool have = false;
    auto replay = [&](int d0, u64 ma, u64 mb) {           // the candidates of anti-diagonals d0 .. d0 + 63, in order
        u64 any = ma | mb;
        while (any) {
            const int t = __ffsll((long long)any) - 1;
            any &= any - 1;
            const int dd = d0 + t;
            if ((ma >> t) & 1) {
                const int j = dd - nrow;
                if (!have || abs(nrow - j) < abs(mi - mj)) { mi = nrow; mj = j; have = true; }
            }
            if ((mb >> t) & 1) {
                const int i = dd - ncol;
                if (!have || mj == ncol || abs(i - ncol) <= abs(mi - mj)) { mi = i; mj = ncol; have = true; }
            }
        }
    };
    constexpr int kEdgeRegs = 8;
    if (nrow + ncol <= 64 * kEdgeRegs) {
        // the usual case (a read against its haplotype window): both edges fit the wavefront's registers -- all loads issued at
        // once, the maximum and the replay from registers (round 3; the two-pass form below paid one memory round trip per 64
        // anti-diagonals in the replay, behind the one for the maximum)
        int va[kEdgeRegs], vb[kEdgeRegs];
#pragma unroll
        for (int q = 0; q < kEdgeRegs; ++q) {
            const int d = 1 + 64 * q + lane, ja = d - nrow, ib = d - ncol;
            va[q] = (use_row && d <= nrow + ncol && ja >= 1) ? last_row[ja] : INT32_MIN;      // (no score is INT32_MIN: MATRIX_MIN_CUTOFF bounds them)
            vb[q] = (d <= nrow + ncol && ib >= 1) ? last_col[ib] : INT32_MIN;
            best = max(best, max(va[q], vb[q]));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) best = max(best, __shfl_xor(best, off, 64));
#pragma unroll
        for (int q = 0; q < kEdgeRegs; ++q) {
            if (1 + 64 * q > nrow + ncol) break;
            replay(1 + 64 * q, __ballot(va[q] == best), __ballot(vb[q] == best));
        }
    } else {
        if (use_row) for (int j = 1 + lane; j <= ncol; j += 64) best = max(best, last_row[j]);
        for (int i = 1 + lane; i <= nrow; i += 64) best = max(best, last_col[i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) best = max(best, __shfl_xor(best, off, 64));
        for (int d0 = 1; d0 <= nrow + ncol; d0 += 64) {
            const int d = d0 + lane;
            const int ja = d - nrow, ib = d - ncol;
            const bool eq_a = use_row && d <= nrow + ncol && ja >= 1 && last_row[ja] == best;
            const bool eq_b = d <= nrow + ncol && ib >= 1 && last_col[ib] == best;
            replay(d0, __ballot(eq_a), __ballot(eq_b));
        }
    }
    // ---- back-trace (PairWiseSW.h:299-408), elements in run-length form
    int i, j;
    if (strategy == MGX_SW_INDEL) { i = nrow; j = ncol; }
    else if (strategy == MGX_SW_LEADING_INDEL) { i = mi; j = ncol; }
    else { i = mi; j = mj; }
    int m = 0;                    // elements written so far
    int cur_op = -1, cur_len = 0; // the open element
    auto flush = [&]() {
        if (cur_op >= 0) { if (lane == 0) { el[2 * m] = (int16_t)cur_op; el[2 * m + 1] = (int16_t)cur_len; } ++m; }
    };
    auto push = [&](int op, int len) {          // append, merging with the open element
        if (op == cur_op) cur_len += len;
        else { flush(); cur_op = op; cur_len = len; }
    };
    if (j < ncol) push(MGX_SW_SOFTCLIP, ncol - j);
    int state = 0;
    int c_i = 0, c_j = 0;         // the cached diagonal starts at (c_i, c_j): lane t holds cell (c_i - t, c_j - t)
    int cached = 0;
    bool valid = false;
    while (i > 0 && j > 0) {
        int btr;
        if (state == 0) {
            int t = c_i - i;
            if (!valid || t < 0 || t >= 64 || c_j - j != t) {
                c_i = i; c_j = j; valid = true; t = 0;
                const int ii = i - lane, jj = j - lane;
                cached = (ii >= 1 && jj >= 1) ? view.cell(ii, jj) : 0;
            }
            const int tt = __builtin_amdgcn_readfirstlane(t);
            // A run of diagonal matches is consumed in ONE step (round 3): the walk is wave-uniform -- one cell per iteration keeps
            // 63 lanes idle for ~275 iterations per pair, and that issue time, not the memory, was most of the trace kernel.  Lanes
            // beyond the matrix hold 0 = a match: the run is bounded by i and j.
            const u64 rest = ~(__ballot((cached & 3) == kOpMatch) >> tt);
            const int run = min(rest ? (int)__builtin_ctzll(rest) : 64, min(i, j));
            if (run > 0) { i -= run; j -= run; push(kOpMatch, run); continue; }
            btr = __builtin_amdgcn_readlane(cached, tt);
        } else {
            btr = view.cell(i, j);
        }
        if (state == kInsertExt) { --j; cur_len++; state = btr & kInsertExt; }
        else if (state == kDeleteExt) { --i; cur_len++; state = btr & kDeleteExt; }
        else {
            const int op = btr & 3;
            if (op == kOpMatch) { --i; --j; push(kOpMatch, 1); state = 0; }
            else if (op == kOpInsert) { --j; push(kOpInsert, 1); state = btr & kInsertExt; }
            else { --i; push(kOpDelete, 1); state = btr & kDeleteExt; }
        }
    }
    int offset;
    if (strategy == MGX_SW_SOFTCLIP) {
        if (j > 0) push(MGX_SW_SOFTCLIP, j);
        offset = i;
    } else if (strategy == MGX_SW_IGNORE) {
        if (j > 0) push(cur_op, j);             // "the last element once more, j long"
        offset = (int16_t)(i - j);
    } else {
        if (i > 0) push(kOpDelete, i);
        else if (j > 0) push(kOpInsert, j);
        offset = 0;
    }
    flush();
    if (lane == 0) {
        SwResult r; r.score = best; r.max_i = mi; r.max_j = mj; r.offset = offset; r.n_elems = m;
        res[J.out_index] = r;
    }
    __syncthreads();                             // lane 0's element stores before the block's lanes copy them
    int16_t* ce = compact + (size_t)J.out_index * 2 * n_compact;
    for (int q = lane; q < 2 * min(m, n_compact); q += 64) ce[q] = el[q];
}

// alignments with more than kCompactElems elements: their element lists are packed densely for one download
struct GatherRef { u64 src, dst; u32 n; u32 pad_; };     // int16 offsets, n = number of int16 values
__global__ __launch_bounds__(256) void k_sw_gather(const GatherRef* __restrict__ refs, u32 n_refs, const int16_t* __restrict__ el,
                                                   int16_t* __restrict__ dense) {
    for (u32 r = blockIdx.x; r < n_refs; r += gridDim.x) {
        const GatherRef g = refs[r];
        for (u32 x = threadIdx.x; x < g.n; x += 256) dense[g.dst + x] = el[g.src + x];
    }
}

// lanes per pair, positions per lane, and which sequence the lanes own
struct Shape { int g, rpl; bool tr; };
// The lanes own the SHORTER sequence (round 3: the read, in the realignment workload) and sweep the longer one: fewer padding
// positions and fewer fill/drain steps per useful step (k_sw_fill<.., TR>).  Up to 256 owned positions two pairs share a
// wavefront (32 lanes x 1..8) whenever the batch is large enough to fill the device that way; otherwise, and beyond, one pair
// per wavefront (64 lanes x 1..32) so that every SIMD still has several wavefronts to hide the dependent integer chain of a
// step behind.  MGX_SW_TRANSPOSE=0 keeps the lanes on the reference (round 2's shapes; A/B and tests).
Shape shape_for(int len1, int len2, bool paired, bool transpose) {
    const bool tr = transpose && len2 < len1;
    const int n_own = tr ? len2 : len1;
    if (paired) {
        static const int cls32[] = {1, 2, 3, 4, 5, 6, 7, 8, 12, 16};
        for (int r : cls32) if (n_own <= 32 * r) return Shape{32, r, tr};
    }
    static const int cls64[] = {1, 2, 3, 4, 5, 6, 8, 16, 32};
    for (int r : cls64) if (n_own <= 64 * r) return Shape{64, r, tr};
    return Shape{0, 0, false};
}
// knobs of one call, read from the environment once (tests and A/B runs set them between calls)
struct Knobs {
    int paired = -1;          // MGX_SW_PAIRED: force two lane groups per wavefront (1) or one (0)
    bool use16 = true;        // MGX_SW_I16=0: the 32-bit fill for every pair
    bool transpose = false;   // MGX_SW_TRANSPOSE=1: 32-bit fill with the lanes over the shorter sequence
    u64 arena = 0;            // MGX_SW_ARENA_LIMIT
    Knobs() {
        if (const char* e = getenv("MGX_SW_PAIRED")) paired = atoi(e) != 0;
        if (const char* e = getenv("MGX_SW_I16")) use16 = atoi(e) != 0;
        if (const char* e = getenv("MGX_SW_TRANSPOSE")) transpose = atoi(e) != 0;
        if (const char* e = getenv("MGX_SW_ARENA_LIMIT")) { const long long v = atoll(e); if (v > 0) arena = (u64)v; }
    }
};
constexpr u32 kPairedFrom = 8192;             // pairs in a chunk from which two lane groups share a wavefront (round 2: 32768)

// ---- the packed 16-bit fill (k_sw_fill16): shapes and admission
Shape shape16_for(int len1, bool paired) {
    static const int cls[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 20, 24, 32};
    if (paired) for (int r : cls) if (len1 <= 32 * r) return Shape{32, r, false};
    for (int r : cls) if (len1 <= 64 * r) return Shape{64, r, false};
    return Shape{0, 0, false};
}
// Every value k_sw_fill16 computes for a cell of the pair's own matrix, and every difference of two of them, must fit 16 bits.
//   H(i, j) >= the all-diagonal path from the boundary: blo + min(len) * min(match, mismatch, 0)
//   any H, E, F <= the best a path can collect: bhi + min(len) * max(match, mismatch, 0) when gaps cost (open, extend <= 0)
//   one more open / extend / match on top of either: `pad`
// LOW_INIT_VALUE only ever meets a real value as LOW + extend against H + open (first column of E, first row of F) and must lose
// strictly: low = L - |extend| - 1.  MATRIX_MIN_CUTOFF (-1e8) cannot bind inside 16 bits.
struct I16Rule {
    int64_t open, extend, dlo, dhi, pos, a_ext, pad;
    bool gaps_cost;
    explicit I16Rule(const SwParams& P) : open(P.open), extend(P.extend), dlo(std::min(P.match, P.mismatch)), dhi(std::max(P.match, P.mismatch)) {
        pos = std::max<int64_t>({dhi, open, extend, 0});
        a_ext = std::llabs((long long)extend);
        pad = std::llabs((long long)open) + a_ext + std::max(std::llabs((long long)P.match), std::llabs((long long)P.mismatch));
        gaps_cost = open <= 0 && extend <= 0;
    }
    bool admits(int64_t n, int64_t m, int32_t* low16) const {
        if (m > 4096) return false;                          // two alternates per lane group are staged in LDS as 32-bit words
        const int64_t mx = std::max(n, m), mn = std::min(n, m);
        const int64_t b_end = open + (mx - 1) * extend;
        const int64_t blo = std::min<int64_t>({0, open, b_end}), bhi = std::max<int64_t>({0, open, b_end});
        const int64_t u0 = gaps_cost ? bhi + mn * std::max<int64_t>(dhi, 0) : bhi + (n + m) * pos;
        const int64_t lh = blo + mn * std::min<int64_t>(dlo, 0);
        const int64_t lo = lh - pad, up = u0 + pad;
        const int64_t low = lo - a_ext - 1, ll = low - a_ext;
        if (ll < -32768 || up > 32767 || up - ll > 32767) return false;
        *low16 = (int32_t)low;
        return true;
    }
};
inline u64 bt_bytes16(u64 steps_max, const Shape& sh) { return ((steps_max + 1) * sh.g * bt_slot16(sh.rpl) + 15) & ~15ull; }
inline u64 lr_bytes16(u64 n_swp, const Shape& sh) { return ((n_swp + 1) * sh.rpl * 4 + 15) & ~15ull; }
constexpr int kCompactElems = 16;             // merged CIGAR elements copied back per pair without a second look
inline u64 bt_bytes(u64 len1, u64 len2, const Shape& sh) {
    const u64 n_own = sh.tr ? len2 : len1, n_swp = sh.tr ? len1 : len2;
    const u64 n_lanes = (n_own + sh.rpl - 1) / sh.rpl, steps = n_swp + n_lanes - 1;
    return ((steps + 1) * sh.g * bt_slot(sh.rpl) + 15) & ~15ull;
}

int itoa_len(int v) { const int neg = v < 0; if (neg) v = -v; int d = 0; while (v > 0) { v /= 10; ++d; } return d + neg; }

// PairWiseSW.h:410-444: last element first; an element whose text does not fit is skipped
int render(const int16_t* el, int n_elems, char* out, int cap) {
    int cur = 0;
    for (int k = n_elems - 1; k >= 0; --k) {
        const int op = el[2 * k], len = el[2 * k + 1];
        const char c = op == kOpMatch ? 'M' : op == kOpInsert ? 'I' : op == kOpDelete ? 'D' : op == MGX_SW_SOFTCLIP ? 'S' : 'R';
        const int need = itoa_len(len) + 1;
        if (need > 1 && cur + need <= cap) {
            int v = len;
            if (v < 0) { out[cur++] = '-'; v = -v; }
            const int d = itoa_len(v);
            for (int q = d - 1; q >= 0; --q) { out[cur + q] = (char)('0' + v % 10); v /= 10; }
            cur += d;
            out[cur++] = c;
        }
    }
    return cur;
}

template <typename T>
struct DevBuf {
    T* p = nullptr; size_t cap = 0;
    int reserve(size_t n) {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const size_t want = n + n / 4 + 64;
        if (hipMalloc((void**)&p, want * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc of %zu bytes failed", want * sizeof(T)); return -ENOMEM; }
        cap = want;
        return 0;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};

}  // namespace

struct mgx_sw {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    DevBuf<u8> d_s1, d_s2, d_bt;
    DevBuf<SwJob> d_jobs;
    DevBuf<int32_t> d_sc;
    DevBuf<int16_t> d_el, d_cel, d_dense;
    DevBuf<GatherRef> d_gather;
    DevBuf<SwResult> d_res;
    mgx_sw_stats_t stats{};
    void* pin = nullptr;          // pinned staging for the uploads (a pageable source is copied at a few GB/s)
    size_t pin_cap = 0;
    void* pin_out = nullptr;      // pinned landing area of the results (a pageable target goes through the runtime's own staging)
    size_t pin_out_cap = 0;
};

namespace {
// back-trace bytes per chunk of a batch (MGX_SW_ARENA_LIMIT overrides).  Measured on 200 000 pairs: with 8-12 GB
// chunks the call took 50-60 ms (a 20 ms stall follows every very large chunk), with 4 GB chunks 29 ms.
constexpr u64 kArenaLimit = 4ull << 30;

template <int G, int RPL>
void launch_fill(mgx_sw* c, const SwJob* jobs, u32 n, u32 max_swept, bool tr, SwParams P) {
    constexpr u32 GPW = 64 / G;
    const u32 stride = (max_swept + 15) & ~15u;
    if (tr) hipLaunchKernelGGL((k_sw_fill<G, RPL, true>), dim3((n + GPW - 1) / GPW), dim3(64), GPW * stride, c->stream, jobs, n, c->d_s1.p, c->d_s2.p,
                               c->d_bt.p, c->d_sc.p, stride, P);
    else    hipLaunchKernelGGL((k_sw_fill<G, RPL, false>), dim3((n + GPW - 1) / GPW), dim3(64), GPW * stride, c->stream, jobs, n, c->d_s1.p, c->d_s2.p,
                               c->d_bt.p, c->d_sc.p, stride, P);
}

// pairs [lo, hi) of the input, already validated
int run_chunk(mgx_sw* c, const mgx_sw_params_t* params, const mgx_sw_input_t* in, u64 lo, u64 hi, const Knobs& knobs,
              int32_t* out_offset, char* out_cigar, u32 stride, int32_t* out_score, int cap_override) {
    const u32 n = (u32)(hi - lo);
    static const bool prof = [] { const char* e = getenv("MGX_SW_PROF"); return e && atoi(e) != 0; }();      // host stage times on stderr
    double tp[6] = {0};
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    tp[0] = now();
    const bool paired = knobs.paired >= 0 ? knobs.paired != 0 : n >= kPairedFrom;
    const SwParams P{params->match, params->mismatch, params->gap_open, params->gap_extend};
    const I16Rule rule(P);
    const u64 base1 = in->ref_off[lo], base2 = in->alt_off[lo];
    const u64 n1 = in->ref_off[hi] - base1, n2 = in->alt_off[hi] - base2;
    // the sequences travel to pinned memory on a helper thread while this one lays out the jobs
    const u32 max_jobs = n + 256;                        // fillers: at most 3 per 16-bit class
    const size_t a1 = (n1 + 255) & ~(size_t)255, a2 = (n2 + 255) & ~(size_t)255, total = a1 + a2 + max_jobs * sizeof(SwJob);
    if (total > c->pin_cap) {
        if (c->pin) (void)hipHostFree(c->pin);
        c->pin = nullptr; c->pin_cap = 0;
        HIP_TRY(hipHostMalloc(&c->pin, total + total / 4, hipHostMallocDefault));
        c->pin_cap = total + total / 4;
    }
    char* const pin = static_cast<char*>(c->pin);
    int rc;
    if ((rc = c->d_s1.reserve(n1 + 16)) || (rc = c->d_s2.reserve(n2 + 16))) return rc;
    hipStream_t s = c->stream;
    std::thread stager[2];
    struct Joiner { std::thread* t; ~Joiner() { for (int k = 0; k < 2; ++k) if (t[k].joinable()) t[k].join(); } } joiner{stager};
    std::atomic<int> stage_err{0};
    const bool staged_apart = n1 + n2 >= (1u << 20);
    if (staged_apart) {
        // each helper also queues its array's upload as soon as it is staged: the copies cross the link while the jobs are laid out
        const int dev = c->device;
        u8* const d1 = c->d_s1.p; u8* const d2 = c->d_s2.p;
        stager[0] = std::thread([=, &stage_err] { memcpy(pin, in->ref + base1, n1);
                                                  if (hipSetDevice(dev) != hipSuccess || hipMemcpyAsync(d1, pin, n1, hipMemcpyHostToDevice, s) != hipSuccess) stage_err = 1; });
        stager[1] = std::thread([=, &stage_err] { memcpy(pin + a1, in->alt + base2, n2);
                                                  if (hipSetDevice(dev) != hipSuccess || hipMemcpyAsync(d2, pin + a1, n2, hipMemcpyHostToDevice, s) != hipSuccess) stage_err = 1; });
    } else { memcpy(pin, in->ref + base1, n1); memcpy(pin + a1, in->alt + base2, n2); }
    // Per pair in input order: class and admission.  Then launch order: the packed 16-bit classes first (32 lanes, then 64; most
    // rows per lane first), the 32-bit classes after them, longest alternate first inside a class (the two pairs of a lane group
    // and the groups of a wavefront finish together; a launch drains on its cheapest jobs).  Two counting sorts: a comparison sort
    // of 10^5 jobs costs more than their alignments.  The score and element arenas are laid out in input order: pair q's score
    // rows start at (bases before it) + 2 q.
    struct Pre { u32 len1, len2; int32_t low16; uint16_t cls; uint8_t rpl, g; };
    std::vector<Pre> pre(n);
    for (u32 q = 0; q < n; ++q) {
        const u64 p = lo + q;
        Pre& J = pre[q];
        J.len1 = (u32)(in->ref_off[p + 1] - in->ref_off[p]); J.len2 = (u32)(in->alt_off[p + 1] - in->alt_off[p]);
        J.low16 = 0;
        if (knobs.use16 && rule.admits(J.len1, J.len2, &J.low16)) {
            const Shape sh = shape16_for((int)J.len1, paired);
            J.rpl = (uint8_t)sh.rpl; J.g = (uint8_t)sh.g;
            J.cls = (uint16_t)((sh.g == 64 ? 64 : 0) + (32 - sh.rpl));                                  // < 128
            c->stats.n_pairs_i16++;
        } else {
            const Shape sh = shape_for((int)J.len1, (int)J.len2, paired, knobs.transpose);
            J.rpl = (uint8_t)sh.rpl; J.g = (uint8_t)sh.g;
            J.cls = (uint16_t)(128 + (sh.g == 64 ? 64 : 0) + (sh.tr ? 128 : 0) + sh.rpl);               // < 384
        }
    }
    std::vector<u32> order(n);
    {
        std::vector<u32> cnt(32770, 0), tmp(n);
        for (u32 q = 0; q < n; ++q) cnt[32767 - pre[q].len2 + 1]++;
        for (u32 k = 0; k < 32768; ++k) cnt[k + 1] += cnt[k];
        for (u32 q = 0; q < n; ++q) tmp[cnt[32767 - pre[q].len2]++] = q;
        u32 c2[385] = {0};
        for (u32 q = 0; q < n; ++q) c2[pre[q].cls + 1]++;
        for (int k = 0; k < 384; ++k) c2[k + 1] += c2[k];
        for (u32 x = 0; x < n; ++x) order[c2[pre[tmp[x]].cls]++] = tmp[x];
    }
    SwJob* const jobs = reinterpret_cast<SwJob*>(pin + a1 + a2);             // built in place, in launch order
    u32 n_jobs = 0;
    auto make_job = [&](u32 q) {
        const u64 p = lo + q;
        const Pre& R = pre[q];
        SwJob J{};
        J.off1 = in->ref_off[p] - base1; J.off2 = in->alt_off[p] - base2;
        J.len1 = R.len1; J.len2 = R.len2; J.rpl = R.rpl; J.strategy = in->strategy[p]; J.out_index = q;
        J.sc_off = J.off1 + J.off2 + 2ull * q; J.el_off = 2 * J.sc_off;
        J.low16 = R.low16;
        J.g = (u32)R.g | (R.cls < 128 ? kJobI16 : (R.cls >= 256 ? kJobTr : 0u));
        return J;
    };
    u64 bt = 0;
    for (u32 x = 0; x < n;) {
        const u32 q0 = order[x];
        if (pre[q0].cls < 128) {
            // one class: lane groups of two pairs, whole wavefronts
            const u32 cl = pre[q0].cls;
            const Shape sh{(int)pre[q0].g, (int)pre[q0].rpl, false};
            const u32 per_wave = 2 * (64 / sh.g);
            u32 y = x;
            while (y < n && pre[order[y]].cls == cl) ++y;
            for (u32 z = x; z < y; z += 2) {
                SwJob A = make_job(order[z]), B{};
                if (z + 1 < y) B = make_job(order[z + 1]);
                else { B.out_index = kNoOutput; B.rpl = A.rpl; B.g = A.g; }
                B.g |= kJobHalf;
                auto lanes = [&](const SwJob& j) { return (u64)(j.len1 + sh.rpl - 1) / sh.rpl; };
                const u64 n_swp = std::max(A.len2, B.len2), steps = n_swp + std::max(lanes(A), lanes(B)) - 1;
                A.bt_off = B.bt_off = bt; bt += bt_bytes16(steps, sh);
                A.lr_off = bt; bt += lr_bytes16(n_swp, sh);
                const bool same_lane = B.len1 > 0 && (A.len1 - 1) / sh.rpl == (B.len1 - 1) / sh.rpl;
                if (same_lane || B.len1 == 0) B.lr_off = A.lr_off;
                else { B.lr_off = bt; bt += lr_bytes16(n_swp, sh); }
                const u64 lc_bytes = (std::max(lanes(A), lanes(B)) * sh.rpl * 4 + 15) & ~15ull;
                A.lc_off = bt; bt += lc_bytes;
                if (B.len2 == A.len2 || B.len1 == 0) B.lc_off = A.lc_off;
                else { B.lc_off = bt; bt += lc_bytes; }
                jobs[n_jobs++] = A; jobs[n_jobs++] = B;
            }
            while (n_jobs % per_wave) {                      // filler groups up to the wavefront boundary
                SwJob F{}; F.out_index = kNoOutput; F.rpl = (u32)sh.rpl; F.g = (u32)sh.g | kJobI16 | (n_jobs % 2 ? kJobHalf : 0u);
                jobs[n_jobs++] = F;
            }
            x = y;
        } else {
            SwJob J = make_job(q0);
            const Shape sh{(int)(J.g & 0xFFu), (int)J.rpl, (J.g & kJobTr) != 0};
            J.bt_off = bt; bt += bt_bytes(J.len1, J.len2, sh);
            jobs[n_jobs++] = J;
            ++x;
        }
    }
    const u64 scn = n1 + n2 + 2ull * n, eln = 2 * scn;
    tp[1] = now();
    if ((rc = c->d_bt.reserve(bt + 16)) || (rc = c->d_jobs.reserve(n_jobs)) ||
        (rc = c->d_sc.reserve(scn)) || (rc = c->d_el.reserve(eln)) || (rc = c->d_res.reserve(n)) ||
        (rc = c->d_cel.reserve((size_t)n * 2 * kCompactElems))) return rc;
    for (auto& t : stager) if (t.joinable()) t.join();
    if (stage_err.load()) { (void)hipGetLastError(); set_error("upload of the sequences failed"); return -EIO; }
    if (!staged_apart) {
        HIP_TRY(hipMemcpyAsync(c->d_s1.p, pin, n1, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->d_s2.p, pin + a1, n2, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(hipMemcpyAsync(c->d_jobs.p, jobs, n_jobs * sizeof(SwJob), hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(c->ev[0], s));
    tp[2] = now();
    for (u32 a = 0; a < n_jobs;) {
        u32 b = a;
        u32 m2 = 0;
        const SwJob* dj = c->d_jobs.p + a;
        if (jobs[a].g & kJobI16) {
            // every class of one lane-group width in one launch (two when classes of more than 16 rows per lane are present)
            const u32 g = jobs[a].g & 0xFFu, gpw = 64 / g;
            const bool big = jobs[a].rpl > 16;
            while (b < n_jobs && (jobs[b].g & kJobI16) && (jobs[b].g & 0xFFu) == g && (jobs[b].rpl > 16) == big) { m2 = std::max(m2, jobs[b].len2); ++b; }
            const u32 stride = (m2 + 15) & ~15u;
            const dim3 grid((b - a) / (2 * gpw));
            const size_t lds = gpw * stride * sizeof(u32);
#define MGX_SW16_LAUNCH(G_, BIG_) hipLaunchKernelGGL((k_sw_fill16<G_, BIG_>), grid, dim3(64), lds, s, dj, b - a, c->d_s1.p, c->d_s2.p, c->d_bt.p, c->d_sc.p, stride, P)
            if (g == 32) { if (big) MGX_SW16_LAUNCH(32, true); else MGX_SW16_LAUNCH(32, false); }
            else         { if (big) MGX_SW16_LAUNCH(64, true); else MGX_SW16_LAUNCH(64, false); }
#undef MGX_SW16_LAUNCH
            c->stats.n_launches++;
            a = b;
            continue;
        }
        const bool tr = (jobs[a].g & kJobTr) != 0;
        while (b < n_jobs && jobs[b].rpl == jobs[a].rpl && jobs[b].g == jobs[a].g) { m2 = std::max(m2, tr ? jobs[b].len1 : jobs[b].len2); ++b; }
#define MGX_SW_CASE(g, r) case r: launch_fill<g, r>(c, dj, b - a, m2, tr, P); break;
        if ((jobs[a].g & 0xFFu) == 32) {
            switch (jobs[a].rpl) {
                MGX_SW_CASE(32, 1) MGX_SW_CASE(32, 2) MGX_SW_CASE(32, 3) MGX_SW_CASE(32, 4) MGX_SW_CASE(32, 5) MGX_SW_CASE(32, 6) MGX_SW_CASE(32, 7)
                MGX_SW_CASE(32, 8) MGX_SW_CASE(32, 12)
                default: launch_fill<32, 16>(c, dj, b - a, m2, tr, P); break;
            }
        } else {
            switch (jobs[a].rpl) {
                MGX_SW_CASE(64, 1) MGX_SW_CASE(64, 2) MGX_SW_CASE(64, 3) MGX_SW_CASE(64, 4) MGX_SW_CASE(64, 5) MGX_SW_CASE(64, 6) MGX_SW_CASE(64, 8)
                MGX_SW_CASE(64, 16)
                default: launch_fill<64, 32>(c, dj, b - a, m2, tr, P); break;
            }
        }
#undef MGX_SW_CASE
        c->stats.n_launches++;
        a = b;
    }
    HIP_TRY(hipEventRecord(c->ev[1], s));
    static const bool lane_trace = [] { const char* e = getenv("MGX_SW_TRACE"); return e && !strcmp(e, "lane"); }();   // A/B: one lane per pair
    if (lane_trace)
        hipLaunchKernelGGL(k_sw_trace, dim3((n_jobs + 63) / 64), dim3(64), 0, s, c->d_jobs.p, n_jobs, c->d_bt.p, c->d_sc.p, c->d_el.p, c->d_res.p,
                           c->d_cel.p, kCompactElems);
    else
        hipLaunchKernelGGL(k_sw_trace_wave, dim3(n_jobs), dim3(64), 0, s, c->d_jobs.p, n_jobs, c->d_bt.p, c->d_sc.p, c->d_el.p, c->d_res.p,
                           c->d_cel.p, kCompactElems);
    HIP_TRY(hipEventRecord(c->ev[2], s));
    HIP_TRY(hipGetLastError());
    tp[3] = now();
    const size_t res_bytes = ((size_t)n * sizeof(SwResult) + 63) & ~(size_t)63, cel_bytes = (size_t)n * 2 * kCompactElems * sizeof(int16_t);
    if (res_bytes + cel_bytes > c->pin_out_cap) {
        if (c->pin_out) (void)hipHostFree(c->pin_out);
        c->pin_out = nullptr; c->pin_out_cap = 0;
        const size_t want = (res_bytes + cel_bytes) * 5 / 4;
        HIP_TRY(hipHostMalloc(&c->pin_out, want, hipHostMallocDefault));
        c->pin_out_cap = want;
    }
    SwResult* const res = static_cast<SwResult*>(c->pin_out);
    int16_t* const cel = reinterpret_cast<int16_t*>(static_cast<char*>(c->pin_out) + res_bytes);
    HIP_TRY(hipMemcpyAsync(res, c->d_res.p, n * sizeof(SwResult), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(cel, c->d_cel.p, cel_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // alignments with more elements than the fixed record holds: one gather launch, one more download
    std::vector<GatherRef> big_refs;
    std::vector<u64> big_at(n, ~0ull);
    {
        u64 eo2 = 0, dense_n = 0;
        for (u32 q = 0; q < n; ++q) {
            const u64 p = lo + q;
            if (res[q].n_elems > kCompactElems) {
                big_at[q] = dense_n;
                big_refs.push_back(GatherRef{eo2, dense_n, (u32)(2 * res[q].n_elems), 0});
                dense_n += 2ull * res[q].n_elems;
            }
            eo2 += 2 * ((in->ref_off[p + 1] - in->ref_off[p]) + (in->alt_off[p + 1] - in->alt_off[p]) + 2);
        }
        if (!big_refs.empty()) {
            if ((rc = c->d_gather.reserve(big_refs.size())) || (rc = c->d_dense.reserve(dense_n))) return rc;
            HIP_TRY(hipMemcpyAsync(c->d_gather.p, big_refs.data(), big_refs.size() * sizeof(GatherRef), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_sw_gather, dim3((u32)std::min<size_t>(big_refs.size(), 65535)), dim3(256), 0, s, c->d_gather.p,
                               (u32)big_refs.size(), c->d_el.p, c->d_dense.p);
        }
        big_at.push_back(dense_n);
    }
    std::unique_ptr<int16_t[]> big(new (std::nothrow) int16_t[big_at.back() + 1]);
    if (!big) return -ENOMEM;
    if (!big_refs.empty()) {
        HIP_TRY(hipMemcpyAsync(big.get(), c->d_dense.p, big_at.back() * sizeof(int16_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1])); c->stats.ms_fill += ms;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[1], c->ev[2])); c->stats.ms_trace += ms;
    c->stats.backtrace_bytes += bt;
    // results and CIGAR text, a few threads when the batch is large (rendering long element lists is
    // the library's biggest host cost)
    tp[4] = now();
    auto emit = [&](u32 q0, u32 q1, u64* cells_out) {
        u64 cells = 0;
        for (u32 q = q0; q < q1; ++q) {
            const u64 p = lo + q;
            const u32 l1 = (u32)(in->ref_off[p + 1] - in->ref_off[p]), l2 = (u32)(in->alt_off[p + 1] - in->alt_off[p]);
            const SwResult& r = res[q];
            out_offset[p] = r.offset;
            if (out_score) out_score[p] = r.score;
            if (out_cigar) {
                char* dst = out_cigar + (size_t)p * stride;
                const int cap = cap_override >= 0 ? cap_override
                                                  : (int)std::min<u64>(2ull * std::max(l1, l2), (u64)stride - 1);   // IntelSmithWaterman.cpp:8
                const int16_t* src = cel + (size_t)q * 2 * kCompactElems;
                if (r.n_elems > kCompactElems) src = big.get() + big_at[q];
                const int len = render(src, r.n_elems, dst, cap);
                dst[len] = 0;                             // cap <= stride - 1
            }
            cells += (u64)l1 * l2;
        }
        *cells_out = cells;
    };
    constexpr u32 kThreads = 4;
    u64 cells[kThreads] = {0};
    if (n < 16384) emit(0, n, &cells[0]);
    else {
        std::thread th[kThreads];
        for (u32 t = 0; t < kThreads; ++t) th[t] = std::thread(emit, (u32)((u64)n * t / kThreads), (u32)((u64)n * (t + 1) / kThreads), &cells[t]);
        for (auto& t : th) t.join();
    }
    for (u64 x : cells) c->stats.cells += x;
    tp[5] = now();
    if (prof) fprintf(stderr, "[mgx_sw] %u pairs: jobs %.3f  stage+H2D %.3f  launches %.3f  kernels+D2H %.3f  text %.3f ms\n", n, tp[1] - tp[0], tp[2] - tp[1],
                      tp[3] - tp[2], tp[4] - tp[3], tp[5] - tp[4]);
    return 0;
}

}  // namespace

extern "C" {

int mgx_sw_create(int device, unsigned flags, mgx_sw_t** out) {
    (void)flags;
    if (!out) { set_error("out is NULL"); return -EINVAL; }
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        set_error("no HIP device is visible (this library has no CPU fallback)");
        return -ENODEV;
    }
    if (device == -1) device = 0;
    if (device < 0 || device >= n_dev) { set_error("device %d out of range", device); return -EINVAL; }
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<mgx_sw> c(new (std::nothrow) mgx_sw);
    if (!c) return -ENOMEM;
    c->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto& e : c->ev) HIP_TRY(hipEventCreate(&e));
    *out = c.release();
    return 0;
}

void mgx_sw_destroy(mgx_sw_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->pin_out) (void)hipHostFree(c->pin_out);
    delete c;
}

static int align_impl(mgx_sw_t* c, const mgx_sw_params_t* params, const mgx_sw_input_t* in,
                      int32_t* out_offset, char* out_cigar, uint32_t cigar_stride, int32_t* out_score, int cap_override) {
    if (!c || !params || !in || !out_offset) { set_error("NULL argument"); return -EINVAL; }
    if (out_cigar && cigar_stride < 2) { set_error("cigar_stride must be at least 2"); return -EINVAL; }
    c->stats = mgx_sw_stats_t{};
    c->stats.n_pairs = in->n_pairs;
    if (in->n_pairs == 0) return 0;
    if (!in->ref_off || !in->alt_off || !in->ref || !in->alt || !in->strategy) { set_error("NULL array in input"); return -EINVAL; }
    if (in->n_pairs > 0x7FFFFFFFull) { set_error("more than 2^31-1 pairs in one batch"); return -E2BIG; }
    for (u64 p = 0; p < in->n_pairs; ++p) {
        if (in->ref_off[p + 1] < in->ref_off[p] || in->alt_off[p + 1] < in->alt_off[p]) { set_error("pair %llu: offsets decrease", (unsigned long long)p); return -EINVAL; }
        const u64 l1 = in->ref_off[p + 1] - in->ref_off[p], l2 = in->alt_off[p + 1] - in->alt_off[p];
        if (l1 == 0 || l2 == 0) { set_error("pair %llu: empty sequence", (unsigned long long)p); return -EINVAL; }
        if (l1 > (u64)kMaxRef) { set_error("pair %llu: reference of %llu bases (limit %d)", (unsigned long long)p, (unsigned long long)l1, kMaxRef); return -E2BIG; }
        if (l2 > 32767) { set_error("pair %llu: alternate of %llu bases (limit 32767)", (unsigned long long)p, (unsigned long long)l2); return -E2BIG; }
        const int st = in->strategy[p];
        if (st < MGX_SW_SOFTCLIP || st > MGX_SW_IGNORE) { set_error("pair %llu: unknown overhang strategy %d", (unsigned long long)p, st); return -EINVAL; }
    }
    HIP_TRY(hipSetDevice(c->device));
    const Knobs knobs;
    const u64 limit = knobs.arena ? knobs.arena : kArenaLimit;
    u64 lo = 0;
    while (lo < in->n_pairs) {
        u64 hi = lo, bt = 0;
        while (hi < in->n_pairs) {
            // an upper bound of what the pair takes of the back-trace arena in any shape: the one-pair-per-wavefront 32-bit form
            // (the paired form and the 16-bit form are smaller) plus the 16-bit kernel's edge rows
            const u64 l1 = in->ref_off[hi + 1] - in->ref_off[hi], l2 = in->alt_off[hi + 1] - in->alt_off[hi];
            const u64 rows = (l1 + 63) / 64, slot = rows <= 8 ? ((rows + 3) & ~3ull) : rows <= 16 ? 16 : 32;
            const u64 swept = knobs.transpose ? std::max(l1, l2) : l2;
            const u64 need = (swept + 64 + 1) * 64 * slot + 8 * (l2 + 1) * ((l1 + 31) / 32 * 3 / 2 + 2) + 8 * (l1 + 64) + 64;
            if (hi > lo && bt + need > limit) break;
            bt += need; ++hi;
        }
        const int rc = run_chunk(c, params, in, lo, hi, knobs, out_offset, out_cigar, cigar_stride, out_score, cap_override);
        if (rc) return rc;
        lo = hi;
    }
    return 0;
}

int mgx_sw_align_batch(mgx_sw_t* c, const mgx_sw_params_t* params, const mgx_sw_input_t* in,
                       int32_t* out_offset, char* out_cigar, uint32_t cigar_stride, int32_t* out_score) {
    return align_impl(c, params, in, out_offset, out_cigar, cigar_stride, out_score, -1);
}

int mgx_sw_align(mgx_sw_t* c, const uint8_t* ref, int refLength, const uint8_t* alt, int altLength,
                 uint8_t* cigar, int cigarLength, int match, int mismatch, int open, int extend, uint8_t strategy) {
    if (!c || !ref || !alt || !cigar || refLength <= 0 || altLength <= 0 || cigarLength < 1) { set_error("bad argument"); return MGX_SW_ALIGN_ERROR(EINVAL); }
    const uint64_t o1[2] = {0, (uint64_t)refLength}, o2[2] = {0, (uint64_t)altLength};
    const mgx_sw_params_t P{match, mismatch, open, extend};
    const mgx_sw_input_t in{1, o1, ref, o2, alt, &strategy};
    int32_t off = 0;
    // the reference renders into a buffer of cigarLength bytes (the caller passes 2 * max(len)), zero-filled
    std::vector<char> tmp((size_t)cigarLength + 1, 0);
    const int rc = align_impl(c, &P, &in, &off, tmp.data(), (uint32_t)cigarLength + 1, nullptr, cigarLength);
    if (rc) return MGX_SW_ALIGN_ERROR(-rc);
    memcpy(cigar, tmp.data(), (size_t)cigarLength);
    return off;
}

int mgx_sw_stats(mgx_sw_t* c, mgx_sw_stats_t* out) {
    if (!c || !out) { set_error("NULL argument"); return -EINVAL; }
    *out = c->stats;
    return 0;
}

}  // extern "C"
