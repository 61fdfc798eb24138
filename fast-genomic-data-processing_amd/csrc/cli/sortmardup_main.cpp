// sortmardup_main.cpp -- `sortmardup`: coordinate sort + mark duplicates, SAM text in, BAM + BAI out.
//
// Same command line and outputs as the reference tool (sortmardup/main.cpp:47-78):
//     sortmardup [-I input.sam] [-t threads] -O output.bam
// text SAM from a file or stdin; output.bam is replaced if it exists; output.bam.bai is written
// next to it; stage timings go to stdout (time_stamp(), main.cpp:597-607).
// Stages: read + parse (threads) -> mgx_sortdedup_pack (host keys, arrival order) ->
// mgx_sortdedup_sort_mark (MI355X: radix sorts + duplicate search) -> BGZF/BAM/BAI (threads).
// There is no CPU fallback: without a HIP device the tool exits with an error.
#include <getopt.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "bam_writer.h"
#include "mgx_pairhmm.h"       // mgx_last_error
#include "mgx_sortdedup.h"
#include "sam_text.h"

namespace {

using clk = std::chrono::steady_clock;
clk::time_point g_t0, g_last;
void time_stamp(const char* hint) {
    const auto now = clk::now();
    printf("%s: %.3f s (total %.3f s)\n", hint, std::chrono::duration<double>(now - g_last).count(),
           std::chrono::duration<double>(now - g_t0).count());
    fflush(stdout);
    g_last = now;
}

bool read_all(const char* path, std::string* out) {
    FILE* f = path ? fopen(path, "rb") : stdin;
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
    if (path) fclose(f);
    return true;
}

struct Parsed {                      // what one parser thread produces for its slice of lines
    std::vector<uint16_t> flag; std::vector<int32_t> tid; std::vector<int64_t> pos;
    std::vector<uint32_t> cigar; std::vector<uint64_t> cigar_len;
    std::vector<uint8_t> qual; std::vector<uint64_t> qual_len;
    std::vector<char> qname; std::vector<uint64_t> qname_len;
    std::vector<uint8_t> blob; std::vector<uint64_t> blob_off;      // BAM bytes per record
    std::vector<int32_t> end;
    std::string err;
};

void parse_slice(const char* data, size_t lo, size_t hi, const samtext::Header& h, Parsed* p) {
    samtext::Record r;
    size_t off = lo;
    while (off < hi) {
        const char* nl = (const char*)memchr(data + off, '\n', hi - off);
        size_t len = nl ? (size_t)(nl - (data + off)) : hi - off;
        const size_t next = off + len + 1;
        if (len && data[off + len - 1] == '\r') --len;
        if (len) {
            if (!samtext::parse_record(data + off, len, h, &r, &p->err)) { p->err += " at: " + std::string(data + off, std::min<size_t>(len, 80)); return; }
            p->flag.push_back(r.flag); p->tid.push_back(r.tid); p->pos.push_back(r.pos);
            p->cigar.insert(p->cigar.end(), r.cigar.begin(), r.cigar.end()); p->cigar_len.push_back(r.cigar.size());
            p->qual.insert(p->qual.end(), r.qual.begin(), r.qual.end()); p->qual_len.push_back(r.qual.size());
            p->qname.insert(p->qname.end(), r.qname.begin(), r.qname.end()); p->qname_len.push_back(r.qname.size());
            p->blob_off.push_back(p->blob.size());
            bamout::encode_record(r, &p->blob);
            p->end.push_back(r.end());
        }
        off = next;
    }
    p->blob_off.push_back(p->blob.size());
}

}  // namespace

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);   // the three sorts overlap on three streams: keep them on distinct hardware queues
    const char* in_path = nullptr; const char* out_path = nullptr;
    int threads = (int)std::thread::hardware_concurrency();
    int device = 0, level = 6;
    int c;
    while ((c = getopt(argc, argv, "I:O:t:d:l:")) >= 0) {
        switch (c) {
            case 'I': in_path = optarg; break;
            case 'O': out_path = optarg; break;
            case 't': threads = atoi(optarg); break;
            case 'd': device = atoi(optarg); break;          // extension: HIP device ordinal
            case 'l': level = atoi(optarg); break;           // extension: deflate level
            default: fprintf(stderr, "usage: %s [-I input.sam] [-t num] -O output.bam\n", argv[0]); return 2;
        }
    }
    if (!out_path) { fprintf(stderr, "usage: %s [-I input.sam] [-t num] -O output.bam\n", argv[0]); return 2; }
    if (threads < 1) threads = 1;
    g_t0 = g_last = clk::now();
    time_stamp("program start");
    unlink(out_path);                                        // main.cpp:66-68

    std::string text;
    if (!read_all(in_path, &text)) { fprintf(stderr, "cannot read %s\n", in_path ? in_path : "stdin"); return 1; }
    samtext::Header hdr;
    const size_t body = samtext::parse_header(text.data(), text.size(), &hdr);
    // ---- parse: slices cut at line boundaries, one parser per thread
    const int T = threads;
    std::vector<size_t> cut(T + 1, text.size());
    cut[0] = body;
    for (int t = 1; t < T; ++t) {
        size_t p = body + (text.size() - body) * (size_t)t / (size_t)T;
        const char* nl = (const char*)memchr(text.data() + p, '\n', text.size() - p);
        cut[t] = nl ? (size_t)(nl - text.data()) + 1 : text.size();
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    std::vector<Parsed> parts(T);
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t) pool.emplace_back(parse_slice, text.data(), cut[t], cut[t + 1], std::cref(hdr), &parts[t]);
        for (auto& th : pool) th.join();
    }
    for (auto& p : parts) if (!p.err.empty()) { fprintf(stderr, "SAM parse error: %s\n", p.err.c_str()); return 1; }
    std::string().swap(text);
    // ---- gather the structure-of-arrays view mgx_sortdedup_pack takes
    size_t n = 0, n_cig = 0, n_q = 0, n_qn = 0;
    for (auto& p : parts) { n += p.flag.size(); n_cig += p.cigar.size(); n_q += p.qual.size(); n_qn += p.qname.size(); }
    std::vector<uint16_t> flag; std::vector<int32_t> tid; std::vector<int64_t> pos; std::vector<int32_t> endv;
    std::vector<uint32_t> cigar; std::vector<uint8_t> qual; std::vector<char> qname;
    std::vector<uint64_t> cigar_off(1, 0), qual_off(1, 0), qname_off(1, 0);
    std::vector<const uint8_t*> blob(n); std::vector<uint32_t> blob_len(n);
    flag.reserve(n); tid.reserve(n); pos.reserve(n); endv.reserve(n); cigar.reserve(n_cig); qual.reserve(n_q); qname.reserve(n_qn);
    size_t k = 0;
    for (auto& p : parts) {
        flag.insert(flag.end(), p.flag.begin(), p.flag.end()); tid.insert(tid.end(), p.tid.begin(), p.tid.end());
        pos.insert(pos.end(), p.pos.begin(), p.pos.end()); endv.insert(endv.end(), p.end.begin(), p.end.end());
        cigar.insert(cigar.end(), p.cigar.begin(), p.cigar.end()); qual.insert(qual.end(), p.qual.begin(), p.qual.end());
        qname.insert(qname.end(), p.qname.begin(), p.qname.end());
        for (size_t i = 0; i < p.flag.size(); ++i, ++k) {
            cigar_off.push_back(cigar_off.back() + p.cigar_len[i]); qual_off.push_back(qual_off.back() + p.qual_len[i]);
            qname_off.push_back(qname_off.back() + p.qname_len[i]);
            blob[k] = p.blob.data() + p.blob_off[i]; blob_len[k] = (uint32_t)(p.blob_off[i + 1] - p.blob_off[i]);
        }
    }
    printf("%zu alignment records, %zu reference sequences\n", n, hdr.ref_name.size());
    time_stamp("read + parse done");

    // ---- keys on the host, sort + duplicate search on the GPU
    mgx_raw_records_t raw{};
    raw.n_records = n; raw.flag = flag.data(); raw.tid = tid.data(); raw.pos = pos.data();
    raw.cigar_off = cigar_off.data(); raw.cigar = cigar.data(); raw.qual_off = qual_off.data(); raw.qual = qual.data();
    raw.qname_off = qname_off.data(); raw.qname = qname.data();
    raw.n_targets = (uint32_t)hdr.ref_len.size(); raw.target_len = hdr.ref_len.data();
    std::vector<mgx_rec_t> recs(n); std::vector<uint32_t> input_index(n); uint64_t L = 0;
    if (mgx_sortdedup_pack(&raw, recs.data(), input_index.data(), &L)) { fprintf(stderr, "pack: %s\n", mgx_last_error()); return 1; }
    time_stamp("pair + key derivation done");
    mgx_sortdedup_t* sd = nullptr;
    if (mgx_sortdedup_create(device, 0, &sd)) { fprintf(stderr, "GPU: %s\n", mgx_last_error()); return 1; }
    std::vector<uint32_t> order(n); std::vector<uint8_t> dup(n);
    if (mgx_sortdedup_sort_mark(sd, L, n, recs.data(), order.data(), dup.data())) { fprintf(stderr, "GPU: %s\n", mgx_last_error()); return 1; }
    mgx_sortdedup_stats_t st{};
    mgx_sortdedup_stats(sd, &st);
    printf("double pairs %llu, single pairs %llu, records marked duplicate %llu, device pipeline %.3f ms\n",
           (unsigned long long)st.n_double, (unsigned long long)st.n_single, (unsigned long long)st.n_dup_records, st.ms_total);
    mgx_sortdedup_destroy(sd);
    time_stamp("sort + duplicate search done");

    // ---- mark + compress + write
    std::vector<bamout::RecordRef> out(n);
    for (size_t q = 0; q < n; ++q) {
        const uint32_t arrival = order[q], src = input_index[arrival];
        out[q] = bamout::RecordRef{blob[src], blob_len[src], tid[src], (int32_t)pos[src], endv[src], dup[arrival] != 0, (flag[src] & 4) == 0};
    }
    std::string err;
    if (!bamout::write_bam(out_path, hdr, out, threads, level, &err)) { fprintf(stderr, "write: %s\n", err.c_str()); return 1; }
    time_stamp("output done");
    return 0;
}
