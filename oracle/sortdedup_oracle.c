/*
 * oracle/sortdedup_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Serial CPU restatement of the sortmardup semantic core (the reference run with one shuffle
 * thread, "-t 1").  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.
 *
 * Parity pin: the reference ships no tests or fixtures for this path and its main() cannot link
 * (it calls five htslib functions that exist nowhere in the tree).  The pin is
 * oracle/_ref/libref_sortdedup.so: the reference's own BAMRecord / SinglePair / DoublePair /
 * bitmap classes compiled in place and driven over the same records
 * (oracle/ref_harness/ref_sortdedup_harness.cpp); tests compare this restatement with it and with
 * the golden file tests/golden/sortdedup_small.npz generated from it.
 *
 * Reference files restated (paths relative to sortmardup/):
 *   tbb/bam_parser.cpp:54-113    ignorable records, mate discovery by adjacent equal qname
 *   tbb/bam_record.cpp:7-62      score(), get_unify_coordinate(), prime5_pos()
 *   tbb/pair.cpp:11-108          qname tile/x/y parse, SinglePair / DoublePair keys
 *   main.cpp:95-108              kTable, reference_length L
 *   main.cpp:145-192             pop loop, arrival order, double_pair_indicator bits
 *   main.cpp:249-281, 299-341    pair sorts and duplicate search
 *   main.cpp:348-357             stable coordinate sort
 *   main.cpp:385-388             BAM_FDUP from duplicate_index
 *
 * Where the reference is undefined (its sort comparator returns true on total equality,
 * main.cpp:264, so the winner among pairs equal in key, score, tile, x and y is whatever
 * std::sort happens to do) this restatement -- and the product -- keep the earliest arrival.
 */
#include <errno.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NO_MATE 0xFFFFFFFFu

typedef struct {
    uint64_t coord, prime5;
    uint32_t mate;
    uint16_t flag, score, tile, x, y, pad_;
} rec_t;   /* identical to mgx_rec_t (include/mgx_sortdedup.h) */

/* ---- B4: bam_record.cpp ----------------------------------------------------------------- */
static uint16_t score_of(const uint8_t* q, uint64_t n) {
    uint16_t r = 0;
    for (uint64_t i = 0; i < n; i++) if (q[i] >= 15) r += q[i];
    return r;
}
static int cigar_type(uint32_t op) { return (0x3C1A7 >> (op << 1)) & 3; }   /* htslib sam.h */

static uint64_t prime5_of(uint64_t coord, int forward, const uint32_t* cig, uint64_t n_cigar) {
    uint64_t tmp = coord;
    if (n_cigar == 0) return tmp;
    if (forward) {
        for (uint64_t i = 0; i < n_cigar; i++) {
            uint32_t op = cig[i] & 15;
            if (op == 4 || op == 5) tmp -= cig[i] >> 4; else return tmp;
        }
        return tmp;
    } else {
        int64_t i = (int64_t)n_cigar - 1;
        while ((cig[i] & 15) == 4 || (cig[i] & 15) == 5) {
            tmp += cig[i] >> 4;
            i--;
            if (i < 0) break;
        }
        for (; i >= 0; i--) if (cigar_type(cig[i] & 15) & 2) tmp += cig[i] >> 4;
        tmp--;
        return tmp;
    }
}

/* ---- B7: pair.cpp:11-49 ------------------------------------------------------------------ */
static uint16_t str_to_u16(const char* s) {
    char* end;
    errno = 0;
    long v = strtol(s, &end, 10);
    return (uint16_t)v;
}
static void tile_x_y(const char* qname, uint64_t len, uint16_t out[3]) {
    char* dup = (char*)malloc(len + 1);
    memcpy(dup, qname, len);
    dup[len] = 0;
    char* save = NULL;
    char* tok[16];
    int n = 0;
    for (char* t = strtok_r(dup, ":", &save); t; t = strtok_r(NULL, ":", &save)) {
        if (n < 16) tok[n] = t;
        n++;
    }
    out[0] = out[1] = out[2] = 0;
    if (n == 7) { out[0] = str_to_u16(tok[4]); out[1] = str_to_u16(tok[5]); out[2] = str_to_u16(tok[6]); }
    else if (n == 6) { out[0] = str_to_u16(tok[3]); out[1] = str_to_u16(tok[4]); out[2] = str_to_u16(tok[5]); }
    free(dup);
}

/* ---- B3 + arrival order: pack raw records ---------------------------------------------------
 * Returns 0.  out_recs / out_input_index have n entries; *out_L = sum of target lengths. */
int sd_oracle_pack(uint64_t n, const uint16_t* flag, const int32_t* tid, const int64_t* pos,
                   const uint64_t* cigar_off, const uint32_t* cigar, const uint64_t* qual_off,
                   const uint8_t* qual, const uint64_t* qname_off, const char* qname,
                   uint32_t n_targets, const uint64_t* target_len, rec_t* out_recs,
                   uint32_t* out_input_index, uint64_t* out_L) {
    uint64_t* ktable = (uint64_t*)malloc(sizeof(uint64_t) * (n_targets + 1));
    uint64_t acc = 0;
    for (uint32_t i = 0; i < n_targets; i++) { ktable[i] = acc; acc += target_len[i]; }
    ktable[n_targets] = acc;
    *out_L = acc;
    uint8_t* popped = (uint8_t*)calloc(n ? n : 1, 1);
    uint64_t k = 0;     /* arrival cursor */
#define IGNORABLE(i) ((flag[i] & (0x4 | 0x100 | 0x800)) != 0)
#define SAME_QNAME(a, b) ((qname_off[(a) + 1] - qname_off[a]) == (qname_off[(b) + 1] - qname_off[b]) && \
                          memcmp(qname + qname_off[a], qname + qname_off[b], qname_off[(a) + 1] - qname_off[a]) == 0)
    for (uint64_t i = 0; i < n; i++) {
        if (popped[i]) continue;
        popped[i] = 1;
        uint64_t mate = (uint64_t)-1;
        if (!IGNORABLE(i)) {
            for (uint64_t q = i + 1; q < n; q++) {
                if (popped[q]) continue;            /* only records still in the queue are seen */
                if (!SAME_QNAME(i, q)) break;
                if (!IGNORABLE(q)) { mate = q; break; }
            }
        }
        uint64_t members[2] = {i, mate};
        int cnt = mate == (uint64_t)-1 ? 1 : 2;
        for (int m = 0; m < cnt; m++) {
            uint64_t r = members[m];
            popped[r] = 1;
            rec_t* o = &out_recs[k + m];
            memset(o, 0, sizeof *o);
            o->coord = tid[r] < 0 ? ktable[n_targets] : ktable[tid[r]] + (uint64_t)pos[r];
            o->prime5 = prime5_of(o->coord, (flag[r] & 0x10) == 0, cigar + cigar_off[r],
                                  cigar_off[r + 1] - cigar_off[r]);
            o->flag = flag[r];
            o->score = score_of(qual + qual_off[r], qual_off[r + 1] - qual_off[r]);
            uint16_t t[3];
            tile_x_y(qname + qname_off[r], qname_off[r + 1] - qname_off[r], t);
            o->tile = t[0]; o->x = t[1]; o->y = t[2];
            o->mate = cnt == 2 ? (uint32_t)(k + (m ^ 1)) : NO_MATE;
            out_input_index[k + m] = (uint32_t)r;
        }
        k += cnt;
    }
    free(popped);
    free(ktable);
    return 0;
}

/* ---- B5, B6, B8: pair entries and their order ----------------------------------------------- */
typedef struct {
    uint64_t key1, key2;      /* sort_key, record2_prime5_pos (0 for singles) */
    uint16_t score, tile, x, y;
    uint32_t rec;             /* arrival index of record1 (= pair identity) */
} pair_t;

static int cmp_pair(const void* a_, const void* b_) {
    const pair_t* a = (const pair_t*)a_; const pair_t* b = (const pair_t*)b_;
    if (a->key1 != b->key1) return a->key1 < b->key1 ? -1 : 1;
    if (a->key2 != b->key2) return a->key2 < b->key2 ? -1 : 1;
    if (a->score != b->score) return a->score > b->score ? -1 : 1;     /* bigger score first */
    if (a->tile != b->tile) return a->tile < b->tile ? -1 : 1;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    if (a->y != b->y) return a->y < b->y ? -1 : 1;
    return a->rec < b->rec ? -1 : (a->rec > b->rec);                    /* total ties: arrival order */
}

typedef struct { uint64_t coord; uint32_t idx; } ckey_t;
static int cmp_coord(const void* a_, const void* b_) {
    const ckey_t* a = (const ckey_t*)a_; const ckey_t* b = (const ckey_t*)b_;
    if (a->coord != b->coord) return a->coord < b->coord ? -1 : 1;
    return a->idx < b->idx ? -1 : (a->idx > b->idx);                    /* stable */
}

static uint64_t g_bits;
/* atomic OR, as the reference's bitmap::set (sortmardup/tbb/bitmap.cpp:20-29): several shards may share one bitmap */
static void bit_set(uint64_t* bm, uint64_t p) { if (p < g_bits) __atomic_fetch_or(&bm[p >> 6], 1ull << (p & 63), __ATOMIC_RELAXED); }
static int bit_get(const uint64_t* bm, uint64_t p) { return p < g_bits ? (int)((bm[p >> 6] >> (p & 63)) & 1) : 0; }

/* out_order[k] = arrival index of the k-th output record; out_dup[i] = 1 iff record i is marked.
 * counts (may be NULL): [0] doubles, [1] singles, [2] duplicate records. */
/* Shard form (one record set over several GPUs): the records that are MARKED (recs, shard-local
 * mate indices) are not the records that are ORDERED (order_coord / order_arrival, n_order entries; NULL = order
 * recs themselves), and `marks` (position << 1 | reverse half) are the ends of pairs living in other shards --
 * what the reference's workers write into the one shared bitmap (main.cpp:181-192). */
/* `shared` (may be NULL): a zeroed bitmap of 4L bits (+ 2 words) owned by the caller and shared by the shards of one
 * record set that run concurrently on the host's cores -- the reference's ONE double_pair_indicator (main.cpp:115).  A
 * bit another shard sets there is a bit this shard's routed marks set anyway, so results do not depend on timing. */
int sd_oracle_run_shard_shared(uint64_t L, uint64_t n, const rec_t* recs, uint64_t n_order, const uint64_t* order_coord,
                        const uint32_t* order_arrival, uint64_t n_marks, const uint64_t* marks, uint32_t* out_order,
                        uint8_t* out_dup, uint64_t* counts, uint64_t* shared) {
    pair_t* dbl = (pair_t*)malloc(sizeof(pair_t) * (n / 2 + 1));
    pair_t* sgl = (pair_t*)malloc(sizeof(pair_t) * (n + 1));
    uint64_t nd = 0, ns = 0;
    /* the reference allocates 4L bits (main.cpp:115) and asserts on anything beyond; positions
     * >= 4L (only reachable through wrapped "negative" 5' ends) are ignored here and in the product */
    const uint64_t maxbit = 4 * L;
    uint64_t* indicator = shared ? shared : (uint64_t*)calloc((maxbit >> 6) + 2, 8);
    g_bits = maxbit;
    memset(out_dup, 0, n);
    for (uint64_t i = 0; i < n_marks; i++) bit_set(indicator, (marks[i] >> 1) + ((marks[i] & 1) ? L : 0));
    for (uint64_t i = 0; i < n; i++) {
        const rec_t* r1 = &recs[i];
        int ign = (r1->flag & (0x4 | 0x100 | 0x800)) != 0;
        if (ign) continue;
        if (r1->mate == NO_MATE) {
            pair_t* p = &sgl[ns++];
            p->key1 = (r1->prime5 << 2) + ((r1->flag & 0x10) ? 3 : 0);   /* pair.cpp:62-68 */
            p->key2 = 0;
            p->score = r1->score; p->tile = r1->tile; p->x = r1->x; p->y = r1->y;
            p->rec = (uint32_t)i;
        } else if (r1->mate > i) {                                        /* record1 of a double pair */
            const rec_t* b1 = r1; const rec_t* b2 = &recs[r1->mate];
            pair_t* p = &dbl[nd++];
            p->score = (uint16_t)(b1->score + b2->score);                 /* pair.cpp:81 */
            p->tile = r1->tile; p->x = r1->x; p->y = r1->y;
            p->rec = (uint32_t)i;
            if (b1->prime5 > b2->prime5) { const rec_t* t = b1; b1 = b2; b2 = t; }   /* :83-85 */
            int f1 = (b1->flag & 0x10) == 0, f2 = (b2->flag & 0x10) == 0;
            int orient = f1 ? (f2 ? 0 : 1) : (f2 ? 2 : 3);                /* FF FR RF RR */
            if (b1->prime5 == b2->prime5 && orient == 2) orient = 1;      /* :102-104 */
            p->key1 = (b1->prime5 << 2) + (uint64_t)orient;
            p->key2 = b2->prime5;
            /* main.cpp:181-192 */
            if (orient == 0 || orient == 2) bit_set(indicator, p->key2); else bit_set(indicator, p->key2 + L);
            if (orient == 0 || orient == 1) bit_set(indicator, p->key1 >> 2); else bit_set(indicator, (p->key1 >> 2) + L);
        }
    }
    qsort(dbl, nd, sizeof(pair_t), cmp_pair);
    for (uint64_t i = 0; i < nd;) {                                        /* main.cpp:269-280 */
        uint64_t j;
        for (j = i + 1; j < nd && dbl[j].key1 == dbl[i].key1 && dbl[j].key2 == dbl[i].key2; j++) {
            out_dup[dbl[j].rec] = 1;
            out_dup[recs[dbl[j].rec].mate] = 1;
        }
        i = j;
    }
    qsort(sgl, ns, sizeof(pair_t), cmp_pair);
    for (uint64_t i = 0; i < ns;) {                                        /* main.cpp:319-340 */
        uint64_t target = sgl[i].key1 >> 2;
        if ((sgl[i].key1 & 3) == 3) target += L;
        if (bit_get(indicator, target)) out_dup[sgl[i].rec] = 1;
        uint64_t j;
        for (j = i + 1; j < ns && sgl[j].key1 == sgl[i].key1; j++) out_dup[sgl[j].rec] = 1;
        i = j;
    }
    const uint64_t no = order_coord ? n_order : n;
    ckey_t* ck = (ckey_t*)malloc(sizeof(ckey_t) * (no + 1));
    for (uint64_t i = 0; i < no; i++) {
        ck[i].coord = order_coord ? order_coord[i] : recs[i].coord;
        ck[i].idx = order_coord ? order_arrival[i] : (uint32_t)i;
    }
    qsort(ck, no, sizeof(ckey_t), cmp_coord);                              /* main.cpp:348-357 */
    uint64_t ndup = 0;
    for (uint64_t i = 0; i < no; i++) out_order[i] = ck[i].idx;
    for (uint64_t i = 0; i < n; i++) ndup += out_dup[i];
    if (counts) { counts[0] = nd; counts[1] = ns; counts[2] = ndup; }
    free(ck); if (!shared) free(indicator); free(dbl); free(sgl);
    return 0;
}

int sd_oracle_run_shard(uint64_t L, uint64_t n, const rec_t* recs, uint64_t n_order, const uint64_t* order_coord,
                        const uint32_t* order_arrival, uint64_t n_marks, const uint64_t* marks, uint32_t* out_order,
                        uint8_t* out_dup, uint64_t* counts) {
    return sd_oracle_run_shard_shared(L, n, recs, n_order, order_coord, order_arrival, n_marks, marks, out_order, out_dup, counts, NULL);
}

int sd_oracle_run(uint64_t L, uint64_t n, const rec_t* recs, uint32_t* out_order, uint8_t* out_dup,
                  uint64_t* counts) {
    return sd_oracle_run_shard(L, n, recs, 0, NULL, NULL, 0, NULL, out_order, out_dup, counts);
}
