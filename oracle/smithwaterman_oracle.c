/*
 * oracle/smithwaterman_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the Smith-Waterman aligner with back-trace that Mutect2Cpp calls right
 * after PairHMM.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker.
 *
 * Parity pin: oracle/_ref/libref_smithwaterman.so -- the reference's own avx2_impl.cc +
 * smithwaterman_common.cc compiled in place (oracle/Makefile `ref`) and driven over the same pairs
 * (tests/test_smithwaterman_oracle.py, golden vectors in tests/golden/smithwaterman.npz).
 *
 * Reference restated (paths relative to deepmutect/Mutect2Cpp-master/src/intel/smithwaterman/):
 *   PairWiseSW.h:31-66     MAIN_CODE: the cell recurrence and its back-trace bits
 *   PairWiseSW.h:70-297    smithWatermanBackTrack: boundaries, anti-diagonal order, best cell
 *   PairWiseSW.h:299-445   getCIGAR: start cell per overhang strategy, state machine, merge, text
 *   smithwaterman_common.h:45-76   op / strategy codes, MATRIX_MIN_CUTOFF, LOW_INIT_VALUE
 *   smithwaterman_common.cc:27-60  fast_itoa
 *
 * The reference computes a whole AVX vector per step, including lanes that fall outside the matrix;
 * those lanes only ever write cells no valid cell reads (rows <= 0, columns > ncol, or the E value of
 * a row that is already finished), so the plain row/column loops below visit the same values.
 * What does depend on the anti-diagonal order is the choice of the best cell among equal scores
 * (PairWiseSW.h:256-285): it is replayed here in that order.
 */
#include <errno.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { OP_MATCH = 0, OP_INSERT = 1, OP_DELETE = 2, BT_INSERT_EXT = 4, BT_DELETE_EXT = 8,
       ST_SOFTCLIP = 9, ST_INDEL = 10, ST_LEADING_INDEL = 11, ST_IGNORE = 12 };
#define MATRIX_MIN_CUTOFF (-100000000)
#define LOW_INIT_VALUE (INT32_MIN / 2)

typedef struct {
    int32_t score, max_i, max_j, offset;
    int32_t n_elems;          /* merged (op, len) elements, in the order getCIGAR holds them (reversed) */
} sw_result_t;

/* DP + back-trace matrix (one byte per cell, row-major (nrow+1) x (ncol+1)) + the best cell. */
static int sw_fill(int32_t match, int32_t mismatch, int32_t open, int32_t extend, const uint8_t* s1, int nrow,
                   const uint8_t* s2, int ncol, int strategy, uint8_t* bt, int32_t* last_row, int32_t* last_col) {
    const int W = ncol + 1;
    int32_t* H0 = malloc(sizeof(int32_t) * (size_t)W);     /* row i-1 */
    int32_t* H1 = malloc(sizeof(int32_t) * (size_t)W);     /* row i   */
    int32_t* F = malloc(sizeof(int32_t) * (size_t)W);      /* per column, PairWiseSW.h:42-49 */
    if (!H0 || !H1 || !F) { free(H0); free(H1); free(F); return -ENOMEM; }
    const int indel = strategy == ST_INDEL || strategy == ST_LEADING_INDEL;
    H0[0] = 0;
    for (int j = 1; j <= ncol; ++j) { H0[j] = indel ? open + (j - 1) * extend : 0; F[j] = LOW_INIT_VALUE; }   /* :243-253 */
    for (int i = 1; i <= nrow; ++i) {
        H1[0] = indel ? open + (i - 1) * extend : 0;
        int32_t E = LOW_INIT_VALUE;                          /* per row, PairWiseSW.h:33-41 */
        for (int j = 1; j <= ncol; ++j) {
            const int32_t open_h = H1[j - 1] + open, ext_h = E + extend;
            E = open_h > ext_h ? open_h : ext_h;
            uint8_t ext = open_h > ext_h ? 0 : BT_INSERT_EXT;
            const int32_t open_v = H0[j] + open, ext_v = F[j] + extend;
            F[j] = ext_v > open_v ? ext_v : open_v;
            if (!(open_v > ext_v)) ext |= BT_DELETE_EXT;
            int32_t h = H0[j - 1] + (s1[i - 1] == s2[j - 1] ? match : mismatch);
            if (h < MATRIX_MIN_CUTOFF) h = MATRIX_MIN_CUTOFF;
            uint8_t b = OP_MATCH;
            if (E > h) { b = OP_INSERT; h = E; }
            if (F[j] > h) { b = OP_DELETE; h = F[j]; }
            H1[j] = h;
            bt[(size_t)i * W + j] = (uint8_t)(b | ext);
            if (i == nrow) last_row[j] = h;
        }
        last_col[i] = H1[ncol];
        int32_t* t = H0; H0 = H1; H1 = t;
    }
    free(H0); free(H1); free(F);
    return 0;
}

/* PairWiseSW.h:256-285 replayed in anti-diagonal order over the last row and the last column */
static void sw_best(const int32_t* last_row, const int32_t* last_col, int nrow, int ncol, int strategy, sw_result_t* r) {
    int32_t best = INT32_MIN, mi = 0, mj = 0;
    for (int d = 1; d <= nrow + ncol; ++d) {
        if (d >= nrow + 1) {                                   /* ilo == nrow + 1: cell (nrow, d - nrow) */
            const int j = d - nrow;
            if (j >= 1 && j <= ncol && (strategy == ST_SOFTCLIP || strategy == ST_IGNORE)) {
                const int32_t s = last_row[j];
                if (best < s || (best == s && abs(nrow - j) < abs(mi - mj))) { best = s; mi = nrow; mj = j; }
            }
        }
        if (d >= ncol + 1) {                                   /* jhi == ncol + 1: cell (d - ncol, ncol) */
            const int i = d - ncol;
            if (i >= 1 && i <= nrow) {
                const int32_t s = last_col[i];
                if (best < s || (best == s && (mj == ncol || abs(i - ncol) <= abs(mi - mj)))) { best = s; mi = i; mj = ncol; }
            }
        }
    }
    r->score = best; r->max_i = mi; r->max_j = mj;
}

/* getCIGAR up to the merged element list (PairWiseSW.h:299-408); elems = (op, len) int16 pairs */
static void sw_trace(const uint8_t* bt, int nrow, int ncol, int strategy, sw_result_t* r, int16_t* elems) {
    const int W = ncol + 1;
    int i, j, n = 0;
    if (strategy == ST_INDEL) { i = nrow; j = ncol; }
    else if (strategy == ST_LEADING_INDEL) { i = r->max_i; j = ncol; }
    else { i = r->max_i; j = r->max_j; }
    if (j < ncol) { elems[2 * n] = ST_SOFTCLIP; elems[2 * n + 1] = (int16_t)(ncol - j); ++n; }
    int state = 0;
    while (i > 0 && j > 0) {
        const int btr = bt[(size_t)i * W + j];
        if (state == BT_INSERT_EXT) { --j; elems[2 * n - 1]++; state = btr & BT_INSERT_EXT; }
        else if (state == BT_DELETE_EXT) { --i; elems[2 * n - 1]++; state = btr & BT_DELETE_EXT; }
        else switch (btr & 3) {
            case OP_MATCH:  --i; --j; elems[2 * n] = OP_MATCH;  elems[2 * n + 1] = 1; state = 0; ++n; break;
            case OP_INSERT: --j;      elems[2 * n] = OP_INSERT; elems[2 * n + 1] = 1; state = btr & BT_INSERT_EXT; ++n; break;
            case OP_DELETE: --i;      elems[2 * n] = OP_DELETE; elems[2 * n + 1] = 1; state = btr & BT_DELETE_EXT; ++n; break;
        }
    }
    if (strategy == ST_SOFTCLIP) {
        if (j > 0) { elems[2 * n] = ST_SOFTCLIP; elems[2 * n + 1] = (int16_t)j; ++n; }
        r->offset = i;
    } else if (strategy == ST_IGNORE) {
        if (j > 0) { elems[2 * n] = elems[2 * (n - 1)]; elems[2 * n + 1] = (int16_t)j; ++n; }
        r->offset = (int16_t)(i - j);
    } else {
        if (i > 0) { elems[2 * n] = OP_DELETE; elems[2 * n + 1] = (int16_t)i; ++n; }
        else if (j > 0) { elems[2 * n] = OP_INSERT; elems[2 * n + 1] = (int16_t)j; ++n; }
        r->offset = 0;
    }
    int m = 0;
    int16_t prev = elems[0];
    for (int k = 1; k < n; ++k) {
        const int16_t cur = elems[2 * k];
        if (cur == prev) elems[2 * m + 1] = (int16_t)(elems[2 * m + 1] + elems[2 * k + 1]);
        else { ++m; elems[2 * m] = cur; elems[2 * m + 1] = elems[2 * k + 1]; prev = cur; }
    }
    r->n_elems = m + 1;
}

static int itoa_len(int32_t v) { int neg = v < 0; if (neg) v = -v; int d = 0; while (v > 0) { v /= 10; ++d; } return d + neg; }

/* PairWiseSW.h:410-444: text, last element first; elements that do not fit are skipped */
int sw_oracle_render(const int16_t* elems, int n_elems, char* out, int cap) {
    int cur = 0;
    for (int k = n_elems - 1; k >= 0; --k) {
        const int op = elems[2 * k], len = elems[2 * k + 1];
        const char c = op == OP_MATCH ? 'M' : op == OP_INSERT ? 'I' : op == OP_DELETE ? 'D' : op == ST_SOFTCLIP ? 'S' : 'R';
        const int need = itoa_len(len) + 1;
        if (need > 1 && cur + need <= cap) {
            char tmp[16]; int v = len, neg = v < 0, d = itoa_len(len), p = 0;
            if (neg) { tmp[p++] = '-'; v = -v; --d; }
            for (int q = d - 1; q >= 0; --q) { tmp[p + q] = (char)('0' + v % 10); v /= 10; }
            memcpy(out + cur, tmp, (size_t)(p + d)); cur += p + d;
            out[cur++] = c;
        }
    }
    return cur;
}

/* One pair.  elems must hold 2*(len1+len2+2) int16; cigar (may be NULL) cap bytes, zero-filled by the caller
 * as IntelSmithWaterman::align does.  Returns 0 or -errno. */
int sw_oracle_align(int32_t match, int32_t mismatch, int32_t open, int32_t extend, const uint8_t* seq1, int len1,
                    const uint8_t* seq2, int len2, int strategy, sw_result_t* res, int16_t* elems, char* cigar, int cigar_cap,
                    int32_t* cigar_len) {
    if (len1 < 0 || len2 < 0) return -EINVAL;
    uint8_t* bt = malloc((size_t)(len1 + 1) * (size_t)(len2 + 1));
    int32_t* lr = malloc(sizeof(int32_t) * (size_t)(len2 + 2));
    int32_t* lc = malloc(sizeof(int32_t) * (size_t)(len1 + 2));
    if (!bt || !lr || !lc) { free(bt); free(lr); free(lc); return -ENOMEM; }
    int rc = sw_fill(match, mismatch, open, extend, seq1, len1, seq2, len2, strategy, bt, lr, lc);
    if (!rc) {
        sw_best(lr, lc, len1, len2, strategy, res);
        sw_trace(bt, len1, len2, strategy, res, elems);
        if (cigar) { const int n = sw_oracle_render(elems, res->n_elems, cigar, cigar_cap); if (cigar_len) *cigar_len = n; }
    }
    free(bt); free(lr); free(lc);
    return rc;
}

/* Batch over concatenated sequences (offset arrays of n+1 entries); elems_off[p] = 2 * (sum of len1+len2+2 before p). */
int sw_oracle_batch(int32_t match, int32_t mismatch, int32_t open, int32_t extend, int n_pairs, const uint64_t* off1,
                    const uint8_t* seq1, const uint64_t* off2, const uint8_t* seq2, const uint8_t* strategy,
                    sw_result_t* res, const uint64_t* elems_off, int16_t* elems) {
    int rc_all = 0;
#pragma omp parallel for schedule(dynamic, 8)
    for (int p = 0; p < n_pairs; ++p) {
        const int rc = sw_oracle_align(match, mismatch, open, extend, seq1 + off1[p], (int)(off1[p + 1] - off1[p]), seq2 + off2[p],
                                       (int)(off2[p + 1] - off2[p]), strategy[p], &res[p], elems + elems_off[p], NULL, 0, NULL);
        if (rc) rc_all = rc;
    }
    return rc_all;
}
