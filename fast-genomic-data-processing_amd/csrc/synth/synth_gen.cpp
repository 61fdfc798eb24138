// synth_gen.cpp -- threaded generators of the deterministic synthetic workloads (bench / test support,
// NOT part of libmgx.so).  Byte-for-byte the streams of synth.py (splitmix64, SURVEY.md section 8d:
// pair i of a workload with seed S draws from the stream whose state starts at S ^ (i * 0xD1B54A32D192ED03)),
// written in C++ so that the 64 M-pair and 200 M-record configurations are generated in seconds.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <unistd.h>
#include <thread>
#include <vector>

namespace {

struct Stream {
    uint64_t s;
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    // synth.SplitMix.bytes(n): ceil(n/8) draws, little-endian bytes, the first n kept
    void bytes(uint8_t* dst, int n) {
        int at = 0;
        for (int d = 0; d < (n + 7) / 8; ++d) {
            const uint64_t z = next();
            const int m = std::min(8, n - at);
            memcpy(dst + at, &z, (size_t)m);
            at += m;
        }
    }
};
constexpr uint64_t kStreamMul = 0xD1B54A32D192ED03ull;

template <class F>
void parallel_for(uint64_t n, int threads, F f) {
    threads = std::max(1, std::min<int>(threads, (int)std::min<uint64_t>(n ? n : 1, 256)));
    if (threads == 1) { f(0, n); return; }
    std::vector<std::thread> th;
    const uint64_t per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        const uint64_t a = std::min(n, t * per), b = std::min(n, a + per);
        if (a < b) th.emplace_back(f, a, b);
    }
    for (auto& x : th) x.join();
}

}  // namespace

extern "C" {

struct synth_pairhmm_params {
    uint64_t seed, first_pair;       // pair indices [first_pair, first_pair + n) of the workload
    int rmin, rmax, hmin, hmax;
    int sub_thr, n_thr;              // int(sub_rate * 65536), int((sub_rate + n_rate) * 65536)
    double random_thr;               // random_read_rate * 65536.0
    int ql, qh, gl, gh, gcp;
    int hap_n_thr;                   // int(hap_n_rate * 256); 0 = no extra draw
};

// pass 1: lengths.  R[n], H[n]
void synth_pairhmm_lengths(const synth_pairhmm_params* p, uint64_t n, int64_t* R, int64_t* H, int threads) {
    parallel_for(n, threads, [=](uint64_t a, uint64_t b) {
        for (uint64_t i = a; i < b; ++i) {
            Stream g{p->seed ^ ((p->first_pair + i) * kStreamMul)};
            const uint64_t w = g.next();
            R[i] = p->rmin + (int64_t)((w & 0xFFFF) % (uint64_t)(p->rmax - p->rmin + 1));
            H[i] = p->hmin + (int64_t)(((w >> 16) & 0xFFFF) % (uint64_t)(p->hmax - p->hmin + 1));
        }
    });
}

// pass 2: bytes.  read_off / hap_off are the prefix sums of R / H ([n + 1])
void synth_pairhmm_fill(const synth_pairhmm_params* p, uint64_t n, const uint64_t* read_off, const uint64_t* hap_off,
                        uint8_t* bases, uint8_t* qual, uint8_t* ins, uint8_t* del, uint8_t* gcp, uint8_t* hap_out, int threads) {
    static const uint8_t ACGT[4] = {'A', 'C', 'G', 'T'};
    parallel_for(n, threads, [=](uint64_t a, uint64_t b) {
        const int rmax = p->rmax, hmax = p->hmax;
        std::vector<uint8_t> hb(hmax + 8), hn(hmax + 8), ev(2 * rmax + 8), rb(rmax + 8), qb(rmax + 8), ib(rmax + 8), db(rmax + 8), hap(hmax + 8);
        for (uint64_t i = a; i < b; ++i) {
            Stream g{p->seed ^ ((p->first_pair + i) * kStreamMul)};
            const uint64_t w = g.next();
            const int64_t R = (int64_t)(read_off[i + 1] - read_off[i]), H = (int64_t)(hap_off[i + 1] - hap_off[i]);
            const int64_t offw = (int64_t)((w >> 32) & 0xFFFF);
            const bool is_random = (double)((w >> 48) & 0xFFFF) < p->random_thr;
            const int64_t span = std::max<int64_t>(H - R, 0);
            const int64_t off = offw % (span + 1);
            g.bytes(hb.data(), hmax);
            for (int c = 0; c < hmax; ++c) hap[c] = ACGT[hb[c] & 3];
            if (p->hap_n_thr > 0) {
                g.bytes(hn.data(), hmax);
                for (int c = 0; c < hmax; ++c) if (hn[c] < p->hap_n_thr) hap[c] = 'N';
            }
            g.bytes(ev.data(), 2 * rmax);
            g.bytes(rb.data(), rmax);
            g.bytes(qb.data(), rmax);
            g.bytes(ib.data(), rmax);
            g.bytes(db.data(), rmax);
            uint8_t* B = bases + read_off[i]; uint8_t* Q = qual + read_off[i]; uint8_t* I = ins + read_off[i];
            uint8_t* D = del + read_off[i]; uint8_t* G = gcp + read_off[i];
            for (int64_t c = 0; c < R; ++c) {
                const uint32_t e16 = (uint32_t)ev[2 * c] | ((uint32_t)ev[2 * c + 1] << 8);
                uint8_t ch = hap[std::min<int64_t>(off + c, hmax - 1)];
                if ((int)e16 < p->sub_thr || is_random) ch = ACGT[rb[c] & 3];
                if ((int)e16 >= p->sub_thr && (int)e16 < p->n_thr) ch = 'N';
                B[c] = ch;
                Q[c] = (uint8_t)(p->ql + qb[c] % (p->qh - p->ql + 1));
                I[c] = (uint8_t)(p->gl + ib[c] % (p->gh - p->gl + 1));
                D[c] = (uint8_t)(p->gl + db[c] % (p->gh - p->gl + 1));
                G[c] = (uint8_t)p->gcp;
            }
            memcpy(hap_out + hap_off[i], hap.data(), (size_t)H);
        }
    });
}


// synth.gen_sortdedup_packed: BASELINE.json configs[3] as packed 32-byte records (mgx_rec_t layout) in arrival
// order; template t fills records 2t and 2t + 1.  out must hold 2 * n_templates records.
struct synth_rec { uint64_t coord, prime5; uint32_t mate; uint16_t flag, score, tile, x, y, pad_; };

void synth_sortdedup_packed(uint64_t seed, uint64_t n_pair_t, uint64_t n_frag_t, uint64_t L, int read_len, double dup_thr,
                            synth_rec* out, int threads) {
    const uint64_t n_t = n_pair_t + n_frag_t;
    struct Tpl { uint64_t start1, start2; bool r1, r2; uint64_t w2; };
    auto base = [=](uint64_t t) {
        Stream g{seed ^ (t * kStreamMul)};
        const uint64_t w = g.next(), w2 = g.next();
        Tpl r;
        r.start1 = w % (L - 4000) + 1000;
        r.start2 = r.start1 + 200 + (w2 & 0xFFFF) % 400;
        const int64_t o = (int64_t)((w2 >> 16) & 0xFF);
        r.r1 = (o >= 230) && ((o < 243) || (o >= 250));
        r.r2 = (o < 230) || (o >= 250);
        r.w2 = w2;
        return r;
    };
    parallel_for(n_t, threads, [=](uint64_t a, uint64_t b) {
        for (uint64_t t = a; t < b; ++t) {
            Tpl me = base(t);
            const bool isdup = t > 0 && (double)((me.w2 >> 24) & 0xFFFF) < dup_thr;
            if (isdup) {
                uint64_t src = (me.w2 >> 40) % (n_t ? n_t : 1);
                src = std::min(src, t - 1);
                const Tpl o = base(src);            // the source's OWN draw (a copy of a copy is not followed)
                me.start1 = o.start1; me.start2 = o.start2; me.r1 = o.r1; me.r2 = o.r2;
            }
            Stream g{seed ^ (t * kStreamMul)};
            g.next(); g.next();
            const uint64_t w3 = g.next();
            const uint64_t clip1 = (w3 & 0xFF) < 26 ? (w3 >> 8) % 20 + 1 : 0;
            const uint64_t clip2 = ((w3 >> 16) & 0xFF) < 26 ? (w3 >> 24) % 20 + 1 : 0;
            const uint64_t sc1 = ((w3 >> 32) & 0xFFFF) % (uint64_t)(read_len * 30) + 1000;
            const uint64_t sc2 = ((w3 >> 48) & 0xFFFF) % (uint64_t)(read_len * 30) + 1000;
            const uint64_t rl = (uint64_t)(read_len - 1);
            synth_rec A{}, B{};
            A.prime5 = me.start1; B.prime5 = me.start2;
            A.coord = me.r1 ? me.start1 - rl + clip1 : me.start1 + clip1;
            B.coord = me.r2 ? me.start2 - rl + clip2 : me.start2 + clip2;
            A.flag = (uint16_t)(1 | 2 | 64 | (me.r1 ? 16 : 0) | (me.r2 ? 32 : 0));
            B.flag = (uint16_t)(1 | 2 | 128 | (me.r2 ? 16 : 0) | (me.r1 ? 32 : 0));
            A.score = (uint16_t)sc1; B.score = (uint16_t)sc2;
            A.tile = B.tile = (uint16_t)((t >> 32) & 0xFFFF); A.x = B.x = (uint16_t)((t >> 16) & 0xFFFF); A.y = B.y = (uint16_t)(t & 0xFFFF);
            A.mate = (uint32_t)(2 * t + 1); B.mate = (uint32_t)(2 * t);
            if (t >= n_pair_t) {                    // fragment: record A mapped and single, B its unmapped mate at the same coordinate
                A.mate = B.mate = 0xFFFFFFFFu;
                A.flag = (uint16_t)((1 | 8 | 64) | (me.r1 ? 16 : 0));
                B.flag = (uint16_t)(1 | 4 | 128);
                B.coord = A.coord; B.prime5 = A.coord;
            }
            out[2 * t] = A; out[2 * t + 1] = B;
        }
    });
}


// SAM text for the packed records of synth_sortdedup_packed (CLI end-to-end runs at scale): one line per record,
// queryname-grouped, a CIGAR whose soft clip separates the coordinate from the 5' end, random bases and qualities.
// Appends to the file `path` (the caller wrote the header); returns the bytes written, or -1.
// (fd >= 0: written to that descriptor instead -- a pipe into the tool's stdin -- and `path` is ignored)
long long synth_sam_text(const synth_rec* recs, uint64_t n, uint64_t first_index, uint64_t contig_len, int n_contigs, int read_len,
                         uint64_t seed, const char* path, int threads, int fd) {
    FILE* f = fd >= 0 ? fdopen(dup(fd), "ab") : fopen(path, "ab");
    if (!f) return -1;
    const uint64_t L = contig_len * (uint64_t)n_contigs;
    long long total = 0;
    const uint64_t block = 1u << 20;                     // records per round: bounded memory
    for (uint64_t b0 = 0; b0 < n; b0 += block) {
        const uint64_t b1 = std::min(n, b0 + block);
        const int T = std::max(1, std::min(threads, 64));
        std::vector<std::string> parts(T);
        const uint64_t per = (b1 - b0 + T - 1) / T;
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
            std::string& out = parts[t];
            const uint64_t a = std::min(b1, b0 + t * per), e = std::min(b1, a + per);
            out.reserve((e - a) * (size_t)(2 * read_len + 120));
            char tmp[256];
            for (uint64_t i = a; i < e; ++i) {
                const synth_rec& r = recs[i];
                const uint64_t tpl = (first_index + i) / 2;
                Stream g{seed ^ ((first_index + i) * kStreamMul)};
                const bool mapped = !(r.flag & 4);
                const uint64_t coord = r.coord < L ? r.coord : 0;
                const int tid = (int)(coord / contig_len);
                const uint64_t pos = coord % contig_len;
                const bool rev = (r.flag & 16) != 0;
                uint64_t clip = 0;
                if (mapped && !rev && r.coord >= r.prime5) clip = std::min<uint64_t>(r.coord - r.prime5, 20);
                int len = snprintf(tmp, sizeof tmp, "SYN:1:FC:1:%u:%u:%u\t%u\tchr%d\t%llu\t%d\t", (unsigned)((tpl >> 32) & 0xFFFF),
                                   (unsigned)((tpl >> 16) & 0xFFFF), (unsigned)(tpl & 0xFFFF), (unsigned)r.flag, tid + 1,
                                   (unsigned long long)(pos + 1), mapped ? 60 : 0);
                out.append(tmp, (size_t)len);
                if (!mapped) out.append("*");
                else if (clip) { len = snprintf(tmp, sizeof tmp, "%lluS%lluM", (unsigned long long)clip, (unsigned long long)(read_len - clip)); out.append(tmp, (size_t)len); }
                else { len = snprintf(tmp, sizeof tmp, "%dM", read_len); out.append(tmp, (size_t)len); }
                len = snprintf(tmp, sizeof tmp, "\t=\t%llu\t%d\t", (unsigned long long)(pos + 1), 0);
                out.append(tmp, (size_t)len);
                const size_t at = out.size();
                out.resize(at + 2 * (size_t)read_len + 2);
                char* p = &out[at];
                static const char ACGT[4] = {'A', 'C', 'G', 'T'};
                for (int c = 0; c < read_len; c += 8) {
                    const uint64_t z = g.next();
                    for (int k = 0; k < 8 && c + k < read_len; ++k) {
                        p[c + k] = ACGT[(z >> (8 * k)) & 3];
                        p[read_len + 1 + c + k] = (char)(33 + 2 + ((z >> (8 * k + 2)) & 63) % 40);
                    }
                }
                p[read_len] = '\t';
                p[2 * read_len + 1] = '\n';
            }
        });
        for (auto& x : th) x.join();
        for (auto& s2 : parts) { if (!s2.empty() && fwrite(s2.data(), 1, s2.size(), f) != s2.size()) { fclose(f); return -1; } total += (long long)s2.size(); }
    }
    fclose(f);
    return total;
}

}  // extern "C"
