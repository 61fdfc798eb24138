// synth_gen.cpp -- threaded generators of the deterministic synthetic workloads (bench / test support,
// NOT part of libmgx.so).  Byte-for-byte the streams of synth.py (splitmix64, SURVEY.md section 8d:
// pair i of a workload with seed S draws from the stream whose state starts at S ^ (i * 0xD1B54A32D192ED03)),
// written in C++ so that the 64 M-pair and 200 M-record configurations are generated in seconds.
#include <algorithm>
#include <cstdint>
#include <cstring>
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

}  // extern "C"
