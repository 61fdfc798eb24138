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
// 64 * RPL consecutive bytes; the trace kernel (one lane per pair) converts (i, j) back to that index.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cerrno>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cstring>
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
    u32 g;               // lanes per pair (32 or 64)
};

struct SwResult { int32_t score, max_i, max_j, offset, n_elems; };
struct SwParams { int match, mismatch, open, extend; };

// G lanes per pair (two pairs share a wavefront when G = 32), RPL rows per lane.  Back-trace byte:
// bits 0-1 op, bit 2 INSERT_EXT, bit 3 DELETE_EXT (PairWiseSW.h:31-66).  (Storing the length of the
// diagonal match run in the spare bits, so that the trace crosses a run in one step, was measured:
// trace 0.58 -> 0.38 ms, fill 1.14 -> 1.42 ms per 20 000 pairs -- not kept.)
template <int G, int RPL>
__global__ __launch_bounds__(64) void k_sw_fill(const SwJob* __restrict__ jobs, u32 n_jobs, const u8* __restrict__ s1, const u8* __restrict__ s2,
                                                u8* __restrict__ bt, int32_t* __restrict__ sc, u32 lds_stride, SwParams P) {
    extern __shared__ u8 sh_all[];                    // the alternate sequences: one global load per step would
    constexpr int GPW = 64 / G;                       // put ~200 serial HBM latencies on every wavefront's path
    const int grp = threadIdx.x / G, lane = threadIdx.x % G;
    const u32 job_idx = blockIdx.x * GPW + grp;
    const bool live = job_idx < n_jobs;
    SwJob J{};
    if (live) J = jobs[job_idx];
    const int nrow = (int)J.len1, ncol = (int)J.len2;
    const bool indel = J.strategy == MGX_SW_INDEL || J.strategy == MGX_SW_LEADING_INDEL;
    const u8* a = s1 + J.off1;
    const u8* b = s2 + J.off2;
    int32_t* last_row = sc + J.sc_off;
    int32_t* last_col = last_row + ncol + 1;
    u8* btp = bt + J.bt_off;
    u8* sh_b = sh_all + (size_t)grp * lds_stride;
    for (int x = lane; x < ncol; x += G) sh_b[x] = b[x];
    __syncthreads();
    const int r0 = lane * RPL;                       // this lane owns rows r0+1 .. r0+RPL (1-based)
    int sa[RPL], hl[RPL], e[RPL];
#pragma unroll
    for (int k = 0; k < RPL; ++k) {
        const int i = r0 + k + 1;
        sa[k] = (live && i <= nrow) ? (int)a[i - 1] : 256;        // padding rows never match and feed nothing above them
        hl[k] = indel ? P.open + (i - 1) * P.extend : 0;          // H(i, 0), PairWiseSW.h:243-253
        e[k] = kLowInit;                                          // E(i, 0)
    }
    int diag_in = r0 == 0 ? 0 : (indel ? P.open + (r0 - 1) * P.extend : 0);   // H(r0, 0)
    const int n_lanes = live ? (nrow + RPL - 1) / RPL : 0;
    const int steps = live ? ncol + n_lanes - 1 : 0;
    int steps_w = steps;                              // the wavefront walks to its longest pair
    if constexpr (GPW > 1) {
        const int other = __shfl_xor(steps, G, 64);
        steps_w = other > steps ? other : steps;
    }
    const int lane_last = live ? (nrow - 1) / RPL : -1, k_last = live ? (nrow - 1) % RPL : 0;
    int send_h = 0, send_f = kLowInit;
    for (int t = 1; t <= steps_w; ++t) {
        // what the lane above computed for this column one step ago: H(r0, j), F(r0, j)
        const int rh = __shfl_up(send_h, 1, G), rf = __shfl_up(send_f, 1, G);
        const int j = t - lane;
        if (j >= 1 && j <= ncol && lane < n_lanes) {
            int up_h, up_f, diag;
            if (lane == 0) {
                up_h = indel ? P.open + (j - 1) * P.extend : 0;                           // H(0, j)
                up_f = kLowInit;                                                          // F(0, j)
                diag = j == 1 ? 0 : (indel ? P.open + (j - 2) * P.extend : 0);            // H(0, j-1), H(0,0) = 0
            } else { up_h = rh; up_f = rf; diag = diag_in; }
            diag_in = up_h;                                          // H(r0, j) is the diagonal of column j+1
            const int c2 = (int)sh_b[j - 1];
            u32 packed[(RPL + 3) / 4];
#pragma unroll
            for (int q = 0; q < (RPL + 3) / 4; ++q) packed[q] = 0;
#pragma unroll
            for (int k = 0; k < RPL; ++k) {
                const int open_h = hl[k] + P.open, ext_h = e[k] + P.extend;
                const int ee = max(open_h, ext_h);
                int ext = open_h > ext_h ? 0 : kInsertExt;
                const int open_v = up_h + P.open, ext_v = up_f + P.extend;
                const int ff = max(ext_v, open_v);
                if (!(open_v > ext_v)) ext |= kDeleteExt;
                int h = max(diag + (sa[k] == c2 ? P.match : P.mismatch), kMinCutoff);
                int op = kOpMatch;
                if (ee > h) { op = kOpInsert; h = ee; }
                if (ff > h) { op = kOpDelete; h = ff; }
                diag = hl[k]; hl[k] = h; e[k] = ee; up_h = h; up_f = ff;
                packed[k >> 2] |= (u32)(op | ext) << ((k & 3) * 8);
            }
            send_h = up_h; send_f = up_f;
            if (lane == lane_last) {
                int v = hl[0];
#pragma unroll
                for (int k = 1; k < RPL; ++k) if (k == k_last) v = hl[k];
                last_row[j] = v;
            }
            u8* dst = btp + ((size_t)t * G + lane) * RPL;
            if constexpr (RPL == 1) dst[0] = (u8)packed[0];
            else if constexpr (RPL == 2) *reinterpret_cast<uint16_t*>(dst) = (uint16_t)packed[0];
            else {
#pragma unroll
                for (int q = 0; q < RPL / 4; ++q) reinterpret_cast<u32*>(dst)[q] = packed[q];
            }
        }
    }
    // a lane's last active step is column ncol: its registers now hold H(i, ncol) (kept out of the loop --
    // a guarded store per cell and step doubled the instruction count of the sweep)
    if (lane < n_lanes) {
#pragma unroll
        for (int k = 0; k < RPL; ++k) if (r0 + k + 1 <= nrow) last_col[r0 + k + 1] = hl[k];
    }
}

// One lane per pair: best end cell, back-trace, merged element list (in the order getCIGAR holds it).
__global__ __launch_bounds__(64) void k_sw_trace(const SwJob* __restrict__ jobs, u32 n, const u8* __restrict__ bt,
                                                 const int32_t* __restrict__ sc, int16_t* __restrict__ elems_all, SwResult* __restrict__ res,
                                                 int16_t* __restrict__ compact, int n_compact) {
    const u32 p = blockIdx.x * 64 + threadIdx.x;
    if (p >= n) return;
    const SwJob J = jobs[p];
    const int nrow = (int)J.len1, ncol = (int)J.len2, strategy = (int)J.strategy, rpl = (int)J.rpl, g = (int)J.g;
    const int32_t* last_row = sc + J.sc_off;
    const int32_t* last_col = last_row + ncol + 1;
    const u8* btp = bt + J.bt_off;
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
        const int l = (i - 1) / rpl, k = (i - 1) - l * rpl;
        const int btr = btp[((size_t)(j + l) * g + l) * rpl + k];
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
    const int nrow = (int)J.len1, ncol = (int)J.len2, strategy = (int)J.strategy, rpl = (int)J.rpl, g = (int)J.g;
    const int32_t* last_row = sc + J.sc_off;
    const int32_t* last_col = last_row + ncol + 1;
    const u8* btp = bt + J.bt_off;
    int16_t* el = elems_all + J.el_off;
    const bool use_row = strategy == MGX_SW_SOFTCLIP || strategy == MGX_SW_IGNORE;
    // ---- best end cell (PairWiseSW.h:256-285).  The maximum first, in parallel; then the cells that reach it are
    //      replayed in anti-diagonal order (row candidate before column candidate), which is all the tie rules see.
    int best = INT32_MIN;
    if (use_row) for (int j = 1 + lane; j <= ncol; j += 64) best = max(best, last_row[j]);
    for (int i = 1 + lane; i <= nrow; i += 64) best = max(best, last_col[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = max(best, __shfl_xor(best, off, 64));
    int mi = 0, mj = 0;
    bool have = false;
    for (int d0 = 1; d0 <= nrow + ncol; d0 += 64) {
        const int d = d0 + lane;
        const int ja = d - nrow, ib = d - ncol;
        const bool eq_a = use_row && d <= nrow + ncol && ja >= 1 && last_row[ja] == best;
        const bool eq_b = d <= nrow + ncol && ib >= 1 && last_col[ib] == best;
        const u64 ma = __ballot(eq_a), mb = __ballot(eq_b);
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
    auto cell_index = [&](int ii, int jj) -> size_t {
        const int l = (ii - 1) / rpl, k = (ii - 1) - l * rpl;
        return ((size_t)(jj + l) * g + l) * rpl + k;
    };
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
                cached = (ii >= 1 && jj >= 1) ? (int)btp[cell_index(ii, jj)] : 0;
            }
            btr = __builtin_amdgcn_readlane(cached, __builtin_amdgcn_readfirstlane(t));
        } else {
            btr = btp[cell_index(i, j)];
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

// lanes per pair and rows per lane: two pairs share a wavefront up to 512 reference bases
struct Shape { int g, rpl; };
// `paired`: batches large enough to fill the device twice over put two pairs on a wavefront (better lane
// use, fewer diagonal fill/drain steps); smaller batches keep one pair per wavefront so that every SIMD
// still has several wavefronts to hide the dependent integer chain of a step behind.
Shape shape_for(int nrow, bool paired) {
    if (paired) {
        static const int cls32[] = {1, 2, 4, 8, 12, 16};
        for (int r : cls32) if (nrow <= 32 * r) return Shape{32, r};
    }
    static const int cls64[] = {1, 2, 4, 8, 16, 32};
    for (int r : cls64) if (nrow <= 64 * r) return Shape{64, r};
    return Shape{0, 0};
}
constexpr u32 kPairedFrom = 32768;            // pairs in a chunk from which two share a wavefront
constexpr int kCompactElems = 16;             // merged CIGAR elements copied back per pair without a second look
inline u64 bt_bytes(u64 len1, u64 len2, const Shape& sh) {
    const u64 n_lanes = (len1 + sh.rpl - 1) / sh.rpl, steps = len2 + n_lanes - 1;
    return ((steps + 1) * sh.g * sh.rpl + 15) & ~15ull;
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
};

namespace {
// back-trace bytes per chunk of a batch (MGX_SW_ARENA_LIMIT overrides).  Measured on 200 000 pairs: with 8-12 GB
// chunks the call took 50-60 ms (a 20 ms stall follows every very large chunk), with 4 GB chunks 29 ms.
constexpr u64 kArenaLimit = 4ull << 30;

template <int G, int RPL>
void launch_fill(mgx_sw* c, const SwJob* jobs, u32 n, u32 max_len2, SwParams P) {
    constexpr u32 GPW = 64 / G;
    const u32 stride = (max_len2 + 15) & ~15u;
    hipLaunchKernelGGL((k_sw_fill<G, RPL>), dim3((n + GPW - 1) / GPW), dim3(64), GPW * stride, c->stream, jobs, n, c->d_s1.p, c->d_s2.p,
                       c->d_bt.p, c->d_sc.p, stride, P);
}

// pairs [lo, hi) of the input, already validated
int run_chunk(mgx_sw* c, const mgx_sw_params_t* params, const mgx_sw_input_t* in, u64 lo, u64 hi,
              int32_t* out_offset, char* out_cigar, u32 stride, int32_t* out_score, int cap_override) {
    const u32 n = (u32)(hi - lo);
    bool paired = n >= kPairedFrom;
    if (const char* e = getenv("MGX_SW_PAIRED")) paired = atoi(e) != 0;      // tests: force either shape family
    std::vector<SwJob> jobs(n);
    const u64 base1 = in->ref_off[lo], base2 = in->alt_off[lo];
    u64 bt = 0, scn = 0, eln = 0;
    for (u32 q = 0; q < n; ++q) {
        const u64 p = lo + q;
        SwJob& J = jobs[q];
        J.off1 = in->ref_off[p] - base1; J.off2 = in->alt_off[p] - base2;
        J.len1 = (u32)(in->ref_off[p + 1] - in->ref_off[p]); J.len2 = (u32)(in->alt_off[p + 1] - in->alt_off[p]);
        const Shape sh = shape_for((int)J.len1, paired);
        J.rpl = (u32)sh.rpl; J.g = (u32)sh.g; J.strategy = in->strategy[p]; J.out_index = q;
        J.bt_off = bt; bt += bt_bytes(J.len1, J.len2, sh);
        J.sc_off = scn; scn += (u64)J.len1 + J.len2 + 2;
        J.el_off = eln; eln += 2 * ((u64)J.len1 + J.len2 + 2);
    }
    // one fill launch per row class: sort the job list by rpl (stable; results go back through out_index)
    {   // counting sort by (g, rpl): a comparison sort of 10^5 jobs costs more than their alignments
        auto cls = [](const SwJob& j) { return (j.g == 64 ? 64u : 0u) + j.rpl; };           // < 128
        u32 cnt[129] = {0};
        for (const SwJob& j : jobs) cnt[cls(j) + 1]++;
        for (int k = 0; k < 128; ++k) cnt[k + 1] += cnt[k];
        std::vector<SwJob> sorted(n);
        for (const SwJob& j : jobs) sorted[cnt[cls(j)]++] = j;
        jobs.swap(sorted);
    }
    const u64 n1 = in->ref_off[hi] - base1, n2 = in->alt_off[hi] - base2;
    int rc;
    if ((rc = c->d_s1.reserve(n1 + 16)) || (rc = c->d_s2.reserve(n2 + 16)) || (rc = c->d_bt.reserve(bt + 16)) || (rc = c->d_jobs.reserve(n)) ||
        (rc = c->d_sc.reserve(scn)) || (rc = c->d_el.reserve(eln)) || (rc = c->d_res.reserve(n)) ||
        (rc = c->d_cel.reserve((size_t)n * 2 * kCompactElems))) return rc;
    hipStream_t s = c->stream;
    {
        const size_t a1 = (n1 + 255) & ~(size_t)255, a2 = (n2 + 255) & ~(size_t)255, total = a1 + a2 + n * sizeof(SwJob);
        if (total > c->pin_cap) {
            if (c->pin) (void)hipHostFree(c->pin);
            c->pin = nullptr; c->pin_cap = 0;
            HIP_TRY(hipHostMalloc(&c->pin, total + total / 4, hipHostMallocDefault));
            c->pin_cap = total + total / 4;
        }
        char* pin = static_cast<char*>(c->pin);
        const char* src[3] = {reinterpret_cast<const char*>(in->ref + base1), reinterpret_cast<const char*>(in->alt + base2),
                              reinterpret_cast<const char*>(jobs.data())};
        char* dst[3] = {pin, pin + a1, pin + a1 + a2};
        const size_t len[3] = {(size_t)n1, (size_t)n2, n * sizeof(SwJob)};
        if (total < (8u << 20)) { for (int k = 0; k < 3; ++k) memcpy(dst[k], src[k], len[k]); }
        else {
            std::thread th[3];
            for (int k = 0; k < 3; ++k) th[k] = std::thread([=] { memcpy(dst[k], src[k], len[k]); });
            for (auto& t : th) t.join();
        }
        HIP_TRY(hipMemcpyAsync(c->d_s1.p, dst[0], n1, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->d_s2.p, dst[1], n2, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->d_jobs.p, dst[2], n * sizeof(SwJob), hipMemcpyHostToDevice, s));
    }
    const SwParams P{params->match, params->mismatch, params->gap_open, params->gap_extend};
    HIP_TRY(hipEventRecord(c->ev[0], s));
    for (u32 a = 0; a < n;) {
        u32 b = a;
        u32 m2 = 0;
        while (b < n && jobs[b].rpl == jobs[a].rpl && jobs[b].g == jobs[a].g) { m2 = std::max(m2, jobs[b].len2); ++b; }
        const SwJob* dj = c->d_jobs.p + a;
        if (jobs[a].g == 32) {
            switch (jobs[a].rpl) {
                case 1: launch_fill<32, 1>(c, dj, b - a, m2, P); break;
                case 2: launch_fill<32, 2>(c, dj, b - a, m2, P); break;
                case 4: launch_fill<32, 4>(c, dj, b - a, m2, P); break;
                case 8: launch_fill<32, 8>(c, dj, b - a, m2, P); break;
                case 12: launch_fill<32, 12>(c, dj, b - a, m2, P); break;
                default: launch_fill<32, 16>(c, dj, b - a, m2, P); break;
            }
        } else {
            switch (jobs[a].rpl) {
                case 1: launch_fill<64, 1>(c, dj, b - a, m2, P); break;
                case 2: launch_fill<64, 2>(c, dj, b - a, m2, P); break;
                case 4: launch_fill<64, 4>(c, dj, b - a, m2, P); break;
                case 8: launch_fill<64, 8>(c, dj, b - a, m2, P); break;
                case 16: launch_fill<64, 16>(c, dj, b - a, m2, P); break;
                default: launch_fill<64, 32>(c, dj, b - a, m2, P); break;
            }
        }
        c->stats.n_launches++;
        a = b;
    }
    HIP_TRY(hipEventRecord(c->ev[1], s));
    static const bool lane_trace = [] { const char* e = getenv("MGX_SW_TRACE"); return e && !strcmp(e, "lane"); }();   // A/B: one lane per pair
    if (lane_trace)
        hipLaunchKernelGGL(k_sw_trace, dim3((n + 63) / 64), dim3(64), 0, s, c->d_jobs.p, n, c->d_bt.p, c->d_sc.p, c->d_el.p, c->d_res.p,
                           c->d_cel.p, kCompactElems);
    else
        hipLaunchKernelGGL(k_sw_trace_wave, dim3(n), dim3(64), 0, s, c->d_jobs.p, n, c->d_bt.p, c->d_sc.p, c->d_el.p, c->d_res.p,
                           c->d_cel.p, kCompactElems);
    HIP_TRY(hipEventRecord(c->ev[2], s));
    HIP_TRY(hipGetLastError());
    std::vector<SwResult> res(n);
    std::unique_ptr<int16_t[]> cel(new (std::nothrow) int16_t[(size_t)n * 2 * kCompactElems]);
    if (!cel) return -ENOMEM;
    HIP_TRY(hipMemcpyAsync(res.data(), c->d_res.p, n * sizeof(SwResult), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(cel.get(), c->d_cel.p, (size_t)n * 2 * kCompactElems * sizeof(int16_t), hipMemcpyDeviceToHost, s));
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
                const int16_t* src = cel.get() + (size_t)q * 2 * kCompactElems;
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
    u64 limit = kArenaLimit;
    if (const char* e = getenv("MGX_SW_ARENA_LIMIT")) { const long long v = atoll(e); if (v > 0) limit = (u64)v; }
    u64 lo = 0;
    while (lo < in->n_pairs) {
        u64 hi = lo, bt = 0;
        while (hi < in->n_pairs) {
            const u64 l1 = in->ref_off[hi + 1] - in->ref_off[hi], l2 = in->alt_off[hi + 1] - in->alt_off[hi];
            const u64 need = bt_bytes(l1, l2, shape_for((int)l1, false)) + 16;     // the one-pair shape is the larger of the two
            if (hi > lo && bt + need > limit) break;
            bt += need; ++hi;
        }
        const int rc = run_chunk(c, params, in, lo, hi, out_offset, out_cigar, cigar_stride, out_score, cap_override);
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
