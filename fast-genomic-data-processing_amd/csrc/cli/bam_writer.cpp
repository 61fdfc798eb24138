// bam_writer.cpp -- see bam_writer.h.  Formats follow SAMv1 sections 4.1 (BGZF), 4.2 (BAM), 5.2 (BAI).
#include "bam_writer.h"

#include "../../../include/mgx_bgzf.h"

#include <fcntl.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

namespace bamout {

std::atomic<int> g_store_arrays_ready{0};

namespace {

constexpr size_t kBlockIn = 0xff00;          // uncompressed bytes per BGZF block
constexpr size_t kBgzfHeader = 18, kBgzfFooter = 8;
constexpr uint32_t kDeviceBatchBlocks = 512;   // BGZF blocks per device batch (32 MB of BAM bytes), two batches per writer thread

template <typename T>
inline void put(std::vector<uint8_t>& a, T v) { const uint8_t* p = (const uint8_t*)&v; a.insert(a.end(), p, p + sizeof(T)); }

// A deflate state that is set up once per writer thread and reset per block (deflateInit2 allocates and clears
// ~256 KB every time; a slice holds thousands of blocks).
struct Deflater {
    z_stream zs; bool ok = false; int level;
    explicit Deflater(int lvl) : level(lvl) {
        memset(&zs, 0, sizeof zs);
        ok = deflateInit2(&zs, lvl, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK;
    }
    ~Deflater() { if (ok) deflateEnd(&zs); }
    Deflater(const Deflater&) = delete;
    Deflater& operator=(const Deflater&) = delete;
};

// one BGZF block for in[0, n); appended to out.  level 0..9
bool bgzf_block(const uint8_t* in, size_t n, Deflater* df, std::vector<uint8_t>* out) {
    uint8_t buf[65536];
    if (!df->ok) return false;
    std::unique_ptr<Deflater> stored;           // level 0, only when the block does not fit compressed (incompressible data)
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (attempt == 1) { stored.reset(new Deflater(0)); if (!stored->ok) return false; }
        z_stream& zs = attempt == 0 ? df->zs : stored->zs;
        if (attempt == 0 && deflateReset(&zs) != Z_OK) return false;
        zs.next_in = const_cast<uint8_t*>(in); zs.avail_in = (uInt)n;
        zs.next_out = buf + kBgzfHeader; zs.avail_out = (uInt)(sizeof buf - kBgzfHeader - kBgzfFooter);
        const int rc = deflate(&zs, Z_FINISH);
        const size_t clen = zs.total_out;
        if (rc != Z_STREAM_END) continue;      // did not fit: retry stored (always fits for n <= 0xff00)
        const size_t total = kBgzfHeader + clen + kBgzfFooter;
        static const uint8_t head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        memcpy(buf, head, 16);
        const uint16_t bsize = (uint16_t)(total - 1);
        memcpy(buf + 16, &bsize, 2);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in, (uInt)n), isize = (uint32_t)n;
        memcpy(buf + kBgzfHeader + clen, &crc, 4);
        memcpy(buf + kBgzfHeader + clen + 4, &isize, 4);
        out->insert(out->end(), buf, buf + total);
        return true;
    }
    return false;
}

struct Slice {                       // output of one writer thread
    std::vector<uint8_t> bytes;      // compressed
    std::vector<uint64_t> vbeg, vend;   // per record: (compressed offset within slice) << 16 | offset in block
    bool ok = true;
};

void compress_slice(const RecordRefs& recs, size_t lo, size_t hi, Deflater* df, Slice* s) {
    std::vector<uint8_t> block;
    block.reserve(kBlockIn + 1024);
    s->vbeg.resize(hi - lo); s->vend.resize(hi - lo);
    auto flush = [&]() {
        if (block.empty()) return;
        if (!bgzf_block(block.data(), block.size(), df, &s->bytes)) s->ok = false;
        block.clear();
    };
    for (size_t k = lo; k < hi; ++k) {
        const RecordRef& r = recs[k];
        const size_t need = 4 + (size_t)r.len;
        if (!block.empty() && block.size() + need > kBlockIn) flush();
        s->vbeg[k - lo] = ((uint64_t)s->bytes.size() << 16) | (uint64_t)block.size();
        const uint32_t bs = r.len;
        put<uint32_t>(block, bs);
        const size_t at = block.size();
        block.insert(block.end(), r.blob, r.blob + r.len);
        if (r.set_dup) { uint16_t f; memcpy(&f, &block[at + kFlagOffset], 2); f |= 0x400; memcpy(&block[at + kFlagOffset], &f, 2); }
        // a record larger than one block spans several: cut greedily
        while (block.size() > kBlockIn) {
            std::vector<uint8_t> rest(block.begin() + kBlockIn, block.end());
            block.resize(kBlockIn);
            flush();
            block = std::move(rest);
        }
        if (block.size() == kBlockIn) { flush(); }
        s->vend[k - lo] = ((uint64_t)s->bytes.size() << 16) | (uint64_t)block.size();
    }
    flush();
}

// The same slice through the device compressor (mgx_bgzf.h): the records are gathered straight into a batch's pinned
// input buffer, block boundaries are only offsets into it, and two batches per writer thread alternate (one being
// filled while the other is on the device).  Virtual offsets are kept as (block ordinal << 16 | offset in block)
// until the compressed position of every block is known.
struct DeviceSlice {
    mgx_bgzf_t* ctx = nullptr;
    mgx_bgzf_batch_t* batch[2] = {nullptr, nullptr};
    uint64_t cap = 0; uint32_t max_blocks = 0;
    std::string err;
    bool create(int device, uint64_t capacity, uint32_t blocks) {
        cap = capacity; max_blocks = blocks;
        if (mgx_bgzf_create(device, 0, &ctx)) { err = mgx_last_error(); return false; }
        for (int i = 0; i < 2; ++i)
            if (mgx_bgzf_batch_create(ctx, cap, max_blocks, &batch[i])) { err = mgx_last_error(); return false; }
        return true;
    }
    ~DeviceSlice() {
        for (int i = 0; i < 2; ++i) if (batch[i]) mgx_bgzf_batch_destroy(ctx, batch[i]);
        if (ctx) mgx_bgzf_destroy(ctx);
    }
};

void compress_slice_device(const RecordRefs& recs, size_t lo, size_t hi, DeviceSlice* dv, Slice* s) {
    s->vbeg.resize(hi - lo); s->vend.resize(hi - lo);
    std::vector<uint64_t> block_at;            // compressed offset of block k of the slice; one more entry = slice length
    int cur = 0;                               // batch being filled
    uint32_t in_flight[2] = {0, 0};            // blocks submitted and not yet collected
    uint8_t* in = mgx_bgzf_batch_input(dv->batch[cur]);
    uint64_t* off = mgx_bgzf_batch_offsets(dv->batch[cur]);
    uint32_t nb = 0;                           // closed blocks of the current batch
    uint64_t fill = 0;                         // bytes in the current batch; the open block is [off[nb], fill)
    uint64_t ordinal = 0;                      // blocks closed so far in the slice
    off[0] = 0;
    auto collect = [&](int k) {
        if (!in_flight[k]) return;
        const uint8_t* o; const uint64_t* oo;
        if (mgx_bgzf_batch_wait(dv->ctx, dv->batch[k], &o, &oo)) { s->ok = false; dv->err = mgx_last_error(); in_flight[k] = 0; return; }
        const uint64_t base = s->bytes.size();
        for (uint32_t i = 0; i < in_flight[k]; ++i) block_at.push_back(base + oo[i]);
        s->bytes.insert(s->bytes.end(), o, o + oo[in_flight[k]]);
        in_flight[k] = 0;
    };
    auto submit = [&]() {                      // sends the closed blocks of the current batch, moves on to the other one
        if (nb == 0) return;
        collect(cur ^ 1);                      // keeps the slice's blocks in order: the older batch first
        if (mgx_bgzf_batch_submit(dv->ctx, dv->batch[cur], nb)) { s->ok = false; dv->err = mgx_last_error(); }
        else in_flight[cur] = nb;
        // the open block's bytes move to the head of the other batch
        const uint64_t open0 = off[nb], open_n = fill - open0;
        uint8_t* in2 = mgx_bgzf_batch_input(dv->batch[cur ^ 1]);
        if (open_n) memcpy(in2, in + open0, open_n);
        cur ^= 1;
        in = in2; off = mgx_bgzf_batch_offsets(dv->batch[cur]);
        off[0] = 0; nb = 0; fill = open_n;
    };
    auto close_block = [&](uint64_t at) {      // the open block ends at byte `at` of the batch
        off[++nb] = at;
        ++ordinal;
    };
    for (size_t k = lo; k < hi && s->ok; ++k) {
        const RecordRef& r = recs[k];
        const uint64_t need = 4 + (uint64_t)r.len;
        if (fill - off[nb] > 0 && fill - off[nb] + need > kBlockIn) close_block(fill);
        // room for the record and for the blocks it may close
        if (fill + need > dv->cap || nb + need / kBlockIn + 2 > dv->max_blocks) {
            if (fill > off[nb] && fill + need > dv->cap && nb == 0) close_block(fill);   // a batch of one short block rather than none
            submit();
            if (fill + need > dv->cap) { s->ok = false; dv->err = "a record larger than the compressor's batch"; break; }
        }
        s->vbeg[k - lo] = (ordinal << 16) | (fill - off[nb]);
        const uint32_t bs = r.len;
        memcpy(in + fill, &bs, 4);
        memcpy(in + fill + 4, r.blob, r.len);
        if (r.set_dup) { uint16_t f; memcpy(&f, in + fill + 4 + kFlagOffset, 2); f |= 0x400; memcpy(in + fill + 4 + kFlagOffset, &f, 2); }
        fill += need;
        while (fill - off[nb] > kBlockIn) close_block(off[nb] + kBlockIn);      // a record larger than one block spans several
        if (fill - off[nb] == kBlockIn) close_block(fill);
        s->vend[k - lo] = (ordinal << 16) | (fill - off[nb]);
    }
    if (fill > off[nb]) close_block(fill);
    submit();
    collect(cur);          // (the batch just left is cur ^ 1 after submit's swap; both are collected, older first)
    collect(cur ^ 1);
    block_at.push_back(s->bytes.size());
    if (s->ok)
        for (size_t i = 0; i < s->vbeg.size(); ++i) {
            s->vbeg[i] = (block_at[s->vbeg[i] >> 16] << 16) | (s->vbeg[i] & 0xffff);
            s->vend[i] = (block_at[s->vend[i] >> 16] << 16) | (s->vend[i] & 0xffff);
        }
}

inline uint64_t rebase(uint64_t v, uint64_t base) { return (((v >> 16) + base) << 16) | (v & 0xffff); }

struct RefIndex {
    std::map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
    std::vector<uint64_t> linear;
    uint64_t off_beg = ~0ull, off_end = 0, n_mapped = 0, n_unmapped = 0;
};

// One contiguous part of the records (output order) into a fresh index: what write_bai's loop does record by record.
template <typename VOff>
void index_part(const RecordRefs& recs, size_t k0, size_t k1, VOff& voff, std::vector<RefIndex>& idx, uint64_t* n_no_coor) {
    int32_t last_tid = -2; uint32_t last_bin = 0; std::vector<std::pair<uint64_t, uint64_t>>* last_chunks = nullptr;
    for (size_t k = k0; k < k1; ++k) {
        const RecordRef& r = recs[k];
        uint64_t vb, ve;
        voff(k, &vb, &ve);
        if (r.tid < 0 || (size_t)r.tid >= idx.size()) { ++*n_no_coor; continue; }
        RefIndex& ri = idx[r.tid];
        const int64_t beg = std::max<int64_t>(r.beg, 0), end = std::max<int64_t>(r.end, beg + 1);
        // the records are coordinate-sorted: neighbours almost always share their bin, so the map is asked once per run
        const uint32_t bin = (uint32_t)reg2bin(beg, end);
        if (r.tid != last_tid || bin != last_bin) { last_chunks = &ri.bins[bin]; last_tid = r.tid; last_bin = bin; }
        auto& chunks = *last_chunks;
        if (!chunks.empty() && chunks.back().second == vb) chunks.back().second = ve;   // contiguous: extend
        else chunks.emplace_back(vb, ve);
        const size_t w0 = (size_t)(beg >> 14), w1 = (size_t)((end - 1) >> 14);
        if (ri.linear.size() <= w1) ri.linear.resize(w1 + 1, 0);
        for (size_t w = w0; w <= w1; ++w) if (ri.linear[w] == 0 || vb < ri.linear[w]) ri.linear[w] = vb;
        ri.off_beg = std::min(ri.off_beg, vb); ri.off_end = std::max(ri.off_end, ve);
        if (r.mapped) ++ri.n_mapped; else ++ri.n_unmapped;
    }
}

// The BAI of the records in output order; voff(k, &vbeg, &vend) gives record k's virtual offsets.  threads > 1 (only for a voff
// that may be called from several threads in any order): the records are indexed in contiguous parts and the parts merged in
// order -- a bin's chunk lists concatenated, the seam joined where the serial loop would have extended a chunk (a part starts
// every bin with a new chunk, which is contiguous with the previous part's last one exactly when the serial rule applies),
// linear windows by minimum, counters by sum: the same bytes as the serial index (200 M records: 1.0 s -> 0.15 s).
template <typename VOff>
bool write_bai(const std::string& path, const samtext::Header& hdr, const RecordRefs& recs, VOff&& voff, std::string* err, int threads = 1) {
    std::vector<RefIndex> idx(hdr.ref_name.size());
    uint64_t n_no_coor = 0;
    size_t T = recs.size() >= (1u << 20) ? (size_t)std::max(1, std::min(threads, 32)) : 1;
    if (threads > 1) if (const char* e = getenv("MGX_CLI_BAI_PARTS")) T = (size_t)std::max(1, std::min(atoi(e), 64));      // tests: parts on small inputs
    T = std::min(T, std::max<size_t>(recs.size(), 1));
    if (T == 1) index_part(recs, 0, recs.size(), voff, idx, &n_no_coor);
    else {
        std::vector<std::vector<RefIndex>> part(T, std::vector<RefIndex>(idx.size()));
        std::vector<uint64_t> part_no(T, 0);
        std::vector<std::thread> gang;
        for (size_t t = 0; t < T; ++t)
            gang.emplace_back([&, t]() { auto v = voff; index_part(recs, recs.size() * t / T, recs.size() * (t + 1) / T, v, part[t], &part_no[t]); });
        for (auto& th : gang) th.join();
        for (size_t t = 0; t < T; ++t) {
            n_no_coor += part_no[t];
            for (size_t r = 0; r < idx.size(); ++r) {
                RefIndex& to = idx[r]; RefIndex& from = part[t][r];
                for (auto& kv : from.bins) {
                    auto& dst = to.bins[kv.first]; auto& src = kv.second;
                    size_t first = 0;
                    if (!dst.empty() && !src.empty() && dst.back().second == src[0].first) { dst.back().second = src[0].second; first = 1; }
                    dst.insert(dst.end(), src.begin() + (std::ptrdiff_t)first, src.end());
                }
                if (to.linear.size() < from.linear.size()) to.linear.resize(from.linear.size(), 0);
                for (size_t w = 0; w < from.linear.size(); ++w)
                    if (from.linear[w] != 0 && (to.linear[w] == 0 || from.linear[w] < to.linear[w])) to.linear[w] = from.linear[w];
                to.off_beg = std::min(to.off_beg, from.off_beg); to.off_end = std::max(to.off_end, from.off_end);
                to.n_mapped += from.n_mapped; to.n_unmapped += from.n_unmapped;
            }
        }
    }
    std::vector<uint8_t> bai;
    bai.insert(bai.end(), {'B', 'A', 'I', 1});
    put<int32_t>(bai, (int32_t)idx.size());
    for (RefIndex& ri : idx) {
        const bool any = !ri.bins.empty();
        put<int32_t>(bai, (int32_t)(ri.bins.size() + (any ? 1 : 0)));
        for (auto& kv : ri.bins) {
            put<uint32_t>(bai, kv.first);
            put<int32_t>(bai, (int32_t)kv.second.size());
            for (auto& c : kv.second) { put<uint64_t>(bai, c.first); put<uint64_t>(bai, c.second); }
        }
        if (any) {                                  // metadata pseudo-bin
            put<uint32_t>(bai, 37450u); put<int32_t>(bai, 2);
            put<uint64_t>(bai, ri.off_beg); put<uint64_t>(bai, ri.off_end);
            put<uint64_t>(bai, ri.n_mapped); put<uint64_t>(bai, ri.n_unmapped);
        }
        // empty windows inherit the previous non-empty offset, as htslib's index does
        uint64_t last = 0;
        for (auto& v : ri.linear) { if (v == 0) v = last; else last = v; }
        put<int32_t>(bai, (int32_t)ri.linear.size());
        for (uint64_t v : ri.linear) put<uint64_t>(bai, v);
    }
    put<uint64_t>(bai, n_no_coor);
    FILE* f = fopen((path + ".bai").c_str(), "wb");
    if (!f) { *err = "cannot open " + path + ".bai"; return false; }
    bool ok = fwrite(bai.data(), 1, bai.size(), f) == bai.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok) { *err = "short write to " + path + ".bai"; return false; }
    return true;
}

// BAM header (magic, text, reference table) as the bytes of the first BGZF blocks
std::vector<uint8_t> header_bytes(const samtext::Header& hdr) {
    std::vector<uint8_t> head;
    head.insert(head.end(), {'B', 'A', 'M', 1});
    put<int32_t>(head, (int32_t)hdr.text.size());
    head.insert(head.end(), hdr.text.begin(), hdr.text.end());
    put<int32_t>(head, (int32_t)hdr.ref_name.size());
    for (size_t i = 0; i < hdr.ref_name.size(); ++i) {
        put<int32_t>(head, (int32_t)hdr.ref_name[i].size() + 1);
        head.insert(head.end(), hdr.ref_name[i].begin(), hdr.ref_name[i].end());
        head.push_back(0);
        put<int32_t>(head, (int32_t)hdr.ref_len[i]);
    }
    return head;
}

bool pwrite_all(int fd, const uint8_t* p, size_t len, uint64_t at) {
    while (len) {
        const ssize_t w = pwrite(fd, p, len, (off_t)at);
        if (w <= 0) { if (w < 0 && errno == EINTR) continue; return false; }
        p += w; len -= (size_t)w; at += (uint64_t)w;
    }
    return true;
}

const uint8_t kEofBlock[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};

}  // namespace

bool write_bam(const std::string& path, const samtext::Header& hdr, const RecordRefs& recs,
               int threads, int level, int device, std::string* err) {
    const bool trace = getenv("MGX_CLI_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what) {
        if (trace) fprintf(stderr, "  write_bam %-28s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    };
    // ---- BAM header, in BGZF blocks of its own
    std::vector<uint8_t> head;
    head.insert(head.end(), {'B', 'A', 'M', 1});
    put<int32_t>(head, (int32_t)hdr.text.size());
    head.insert(head.end(), hdr.text.begin(), hdr.text.end());
    put<int32_t>(head, (int32_t)hdr.ref_name.size());
    for (size_t i = 0; i < hdr.ref_name.size(); ++i) {
        put<int32_t>(head, (int32_t)hdr.ref_name[i].size() + 1);
        head.insert(head.end(), hdr.ref_name[i].begin(), hdr.ref_name[i].end());
        head.push_back(0);
        put<int32_t>(head, (int32_t)hdr.ref_len[i]);
    }
    std::vector<uint8_t> file;
    if (device >= 0) {
        mgx_bgzf_t* ctx = nullptr;
        if (mgx_bgzf_create(device, 0, &ctx)) { *err = mgx_last_error(); return false; }
        std::vector<uint64_t> off;
        for (size_t o = 0; o < head.size(); o += kBlockIn) off.push_back(o);
        off.push_back(head.size());
        std::vector<uint64_t> out_off(off.size());
        file.resize(mgx_bgzf_bound(head.size(), off.size() - 1));
        const int rc = mgx_bgzf_compress(ctx, head.data(), off.data(), off.size() - 1, file.data(), file.size(), out_off.data());
        mgx_bgzf_destroy(ctx);
        if (rc) { *err = mgx_last_error(); return false; }
        file.resize(out_off.back());
    } else {
        Deflater df(level);
        for (size_t off = 0; off < head.size(); off += kBlockIn)
            if (!bgzf_block(head.data() + off, std::min(kBlockIn, head.size() - off), &df, &file)) { *err = "deflate failed"; return false; }
    }
    const uint64_t header_end = file.size();
    stamp("header compressed");

    // ---- records: contiguous slices compressed independently (sortmardup/main.cpp:371-421)
    const size_t n = recs.size();
    // device compressor: the writer threads only gather records into pinned memory, a few of them saturate it
    const int T = device >= 0 ? std::max(1, std::min(threads, 8)) : std::max(1, threads);
    const size_t n_slices = n ? std::min<size_t>((size_t)T * (device >= 0 ? 16 : 4), (n + 4095) / 4096) : 0;
    std::vector<Slice> slices(n_slices);
    std::vector<size_t> lo(n_slices + 1, 0);
    for (size_t s = 0; s <= n_slices; ++s) lo[s] = n_slices ? n * s / n_slices : 0;
    // the file is written while the slices are still being compressed: as soon as slice s is done its place in the file
    // is known, and one of a few writer threads puts it there (pwrite) and drops its bytes; the pool compresses ahead
    const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { *err = "cannot open " + path; return false; }
    auto put_at = [fd](const uint8_t* p, size_t len, uint64_t at) -> bool {
        while (len) {
            const ssize_t w = pwrite(fd, p, len, (off_t)at);
            if (w <= 0) { if (w < 0 && errno == EINTR) continue; return false; }
            p += w; len -= (size_t)w; at += (uint64_t)w;
        }
        return true;
    };
    std::atomic<bool> write_ok{put_at(file.data(), file.size(), 0)};
    std::vector<uint64_t> base(n_slices + 1, header_end);
    std::string dev_err_copy;
    {
        std::vector<std::thread> pool;
        size_t next = 0;
        std::mutex mu;
        std::condition_variable cv;
        std::vector<char> done(n_slices, 0);
        std::string dev_err;
        for (int t = 0; t < T; ++t)
            pool.emplace_back([&]() {
                std::unique_ptr<Deflater> df;
                std::unique_ptr<DeviceSlice> dv;
                bool dev_ok = true;
                if (device >= 0) {
                    dv.reset(new DeviceSlice);
                    dev_ok = dv->create(device, (uint64_t)kDeviceBatchBlocks * kBlockIn, kDeviceBatchBlocks);
                    stamp("a writer's batches are ready");
                } else {
                    df.reset(new Deflater(level));
                }
                for (;;) {
                    size_t s;
                    { std::lock_guard<std::mutex> g(mu); if (next >= n_slices) return; s = next++; }
                    if (device >= 0) {
                        if (dev_ok) compress_slice_device(recs, lo[s], lo[s + 1], dv.get(), &slices[s]); else slices[s].ok = false;
                        if (!slices[s].ok) { std::lock_guard<std::mutex> g(mu); if (dev_err.empty()) dev_err = dv->err; }
                    } else {
                        compress_slice(recs, lo[s], lo[s + 1], df.get(), &slices[s]);
                    }
                    { std::lock_guard<std::mutex> g(mu); done[s] = 1; }
                    cv.notify_all();
                }
            });
        std::vector<std::thread> writers;
        std::vector<size_t> wq;                 // slices whose place is known, not yet written
        size_t wq_next = 0; bool wq_closed = false;
        std::mutex wmu; std::condition_variable wcv;
        // ONE writer: buffered pwrite()s of one file take its inode lock in turn, and threads that queue for it write slower than
        // a single one does (tools/dev_tmpfs_write.py on the GPU box: 1 thread 6.3 GB/s, 4 threads 3.7, 8 threads 3.5)
        for (int t = 0; t < 1; ++t)
            writers.emplace_back([&]() {
                for (;;) {
                    size_t s;
                    {
                        std::unique_lock<std::mutex> lk(wmu);
                        wcv.wait(lk, [&] { return wq_next < wq.size() || wq_closed; });
                        if (wq_next >= wq.size()) return;
                        s = wq[wq_next++];
                    }
                    if (slices[s].ok && !put_at(slices[s].bytes.data(), slices[s].bytes.size(), base[s])) write_ok = false;
                    std::vector<uint8_t>().swap(slices[s].bytes);
                }
            });
        for (size_t s = 0; s < n_slices; ++s) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done[s] != 0; }); }
            base[s + 1] = base[s] + slices[s].bytes.size();
            if (s == 0) stamp("first slice compressed");
            { std::lock_guard<std::mutex> g(wmu); wq.push_back(s); }
            wcv.notify_one();
        }
        { std::lock_guard<std::mutex> g(wmu); wq_closed = true; }
        wcv.notify_all();
        stamp("last slice compressed");
        for (auto& th : pool) th.join();
        stamp("compressors released");
        for (auto& th : writers) th.join();
        stamp("file written");
        dev_err_copy = dev_err;
    }
    for (size_t s = 0; s < n_slices; ++s)
        if (!slices[s].ok) { close(fd); *err = device >= 0 ? "device compressor: " + dev_err_copy : std::string("deflate failed"); return false; }
    static const uint8_t eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool ok = write_ok.load() && put_at(eof_block, 28, base[n_slices]);
    ok = (close(fd) == 0) && ok;
    if (!ok) { *err = "short write to " + path; return false; }

    // ---- BAI (merge of the per-slice offsets, the job of the reference's merge_index)
    size_t cur_slice = 0;
    auto voff = [&](size_t k, uint64_t* vb, uint64_t* ve) {
        while (k >= lo[cur_slice + 1]) ++cur_slice;          // k only grows
        *vb = rebase(slices[cur_slice].vbeg[k - lo[cur_slice]], base[cur_slice]);
        *ve = rebase(slices[cur_slice].vend[k - lo[cur_slice]], base[cur_slice]);
    };
    const bool iok = write_bai(path, hdr, recs, voff, err);
    stamp("index written");
    return iok;
}


// The records live in HBM (mgx_bgzf_store): recs[k].blob is a DEVICE address.  The device gathers them in output order,
// marks the duplicates, cuts the stream every 65 280 bytes and compresses it; this thread writes the blocks as they come
// back and keeps the compressed position of every block for the index.
bool write_bam_store(const std::string& path, const samtext::Header& hdr, const RecordRefs& recs, void* bgzf_ctx, void* store,
                     int threads, std::string* err) {
    const bool trace = getenv("MGX_CLI_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto stamp = [&](const char* what) {
        if (trace) fprintf(stderr, "  write_bam_store %-22s %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
    };
    mgx_bgzf_t* ctx = (mgx_bgzf_t*)bgzf_ctx;
    const std::vector<uint8_t> head = header_bytes(hdr);
    std::vector<uint8_t> file;
    {
        std::vector<uint64_t> off;
        for (size_t o = 0; o < head.size(); o += kBlockIn) off.push_back(o);
        off.push_back(head.size());
        std::vector<uint64_t> out_off(off.size());
        file.resize(mgx_bgzf_bound(head.size(), off.size() - 1));
        if (mgx_bgzf_compress(ctx, head.data(), off.data(), off.size() - 1, file.data(), file.size(), out_off.data())) { *err = mgx_last_error(); return false; }
        file.resize(out_off.back());
    }
    const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { *err = "cannot open " + path; return false; }
    bool ok = pwrite_all(fd, file.data(), file.size(), 0);
    const size_t n = recs.size();
    // (arrays of 200 M entries: left uninitialised, so that their pages are first touched by the gang below and not zeroed by one thread)
    std::unique_ptr<uint32_t[]> order(new uint32_t[n + 1]), len(new uint32_t[n + 1]);
    std::unique_ptr<uint8_t[]> dup(new uint8_t[n + 1]);
    std::unique_ptr<uint64_t[]> addr(new uint64_t[n + 1]), uoff(new uint64_t[n + 1]);
    {
        std::vector<std::thread> gang;
        const size_t T = (size_t)std::max(1, std::min(threads, 16));
        for (size_t t = 0; t < T; ++t)
            gang.emplace_back([&, t]() {
                for (size_t q = n * t / T, e = n * (t + 1) / T; q < e; ++q) {
                    order[q] = (uint32_t)q; len[q] = recs[q].len; dup[q] = recs[q].set_dup; addr[q] = (uint64_t)(uintptr_t)recs[q].blob;
                }
            });
        for (auto& th : gang) th.join();
    }
    g_store_arrays_ready.store(1);
    stamp("arrays ready");
    // The sink must be done with a batch's bytes when it returns.  Round 3: it simply writes them -- one pwrite of the whole batch
    // from the pinned buffer, on the thread store_emit calls it on.  The device does not wait for that (store_emit keeps three
    // windows in flight, and a window takes the device 1.5 ms against 7 ms in the file), and ONE writer is what the file takes
    // fastest: buffered pwrite()s of one file go through its inode lock one at a time, and threads queueing for it move less than
    // a single one (tools/dev_tmpfs_write.py on the GPU box, 16 GB into tmpfs: 1 thread 6.3 GB/s; 4 threads on one file 3.7;
    // 8 threads 3.5; a file each 18.7 / 33.5 -- a BAM is one file).  Rounds 1-2 had four helpers copy a quarter of the batch each
    // and write it in the background: 3.5 GB/s at 200 M records.  MGX_CLI_WRITERS=n (n > 0) brings the helpers back.
    struct Helper {
        std::thread th; std::mutex mu; std::condition_variable cv;
        const uint8_t* src = nullptr; uint64_t n = 0, at = 0; int state = 0;      // 0 idle, 1 job posted, 2 copied (writing), -1 quit
        std::vector<uint8_t> buf; bool ok = true;
    };
    struct Sink { int fd; uint64_t at; std::vector<uint64_t> block_at; std::vector<Helper> h; } sk;
    sk.fd = fd; sk.at = file.size();
    { const char* e = getenv("MGX_CLI_WRITERS"); const int nh = e ? atoi(e) : 0; sk.h = std::vector<Helper>((size_t)std::max(0, std::min(nh, 32))); }
    for (Helper& h : sk.h)
        h.th = std::thread([&h, fd]() {
            for (;;) {
                std::unique_lock<std::mutex> lk(h.mu);
                h.cv.wait(lk, [&] { return h.state == 1 || h.state == -1; });
                if (h.state == -1) return;
                if (h.buf.size() < h.n) h.buf.resize(h.n);
                memcpy(h.buf.data(), h.src, h.n);
                h.state = 2;
                h.cv.notify_all();
                lk.unlock();
                const bool w = pwrite_all(fd, h.buf.data(), h.n, h.at);
                lk.lock();
                if (!w) h.ok = false;
                h.state = 0;
                h.cv.notify_all();
            }
        });
    auto sink = [](void* user, const uint8_t* blocks, uint64_t n_bytes, uint32_t n_blocks, const uint64_t* block_off) -> int {
        Sink* k = (Sink*)user;
        for (uint32_t i = 0; i < n_blocks; ++i) k->block_at.push_back(k->at + block_off[i]);
        const int nh = (int)k->h.size();
        if (nh == 0) {
            if (!pwrite_all(k->fd, blocks, n_bytes, k->at)) return 1;
            k->at += n_bytes;
            return 0;
        }
        for (int t = 0; t < nh; ++t) {
            Helper& h = k->h[t];
            const uint64_t lo = n_bytes * t / nh, hi = n_bytes * (t + 1) / nh;
            std::unique_lock<std::mutex> lk(h.mu);
            h.cv.wait(lk, [&] { return h.state == 0; });          // its previous quarter is on disk
            h.src = blocks + lo; h.n = hi - lo; h.at = k->at + lo; h.state = 1;
            h.cv.notify_all();
        }
        bool ok = true;
        for (int t = 0; t < nh; ++t) {
            Helper& h = k->h[t];
            std::unique_lock<std::mutex> lk(h.mu);
            h.cv.wait(lk, [&] { return h.state != 1; });          // copied: the batch's buffer may be reused
            ok = ok && h.ok;
        }
        if (!ok) return 1;
        k->at += n_bytes;
        return 0;
    };
    const int emit_rc = mgx_bgzf_store_emit((mgx_bgzf_store_t*)store, n, order.get(), dup.get(), addr.get(), len.get(), sink, &sk, uoff.get());
    bool helpers_ok = true;
    for (Helper& h : sk.h) {
        { std::unique_lock<std::mutex> lk(h.mu); h.cv.wait(lk, [&] { return h.state == 0; }); h.state = -1; h.cv.notify_all(); }
        h.th.join();
        helpers_ok = helpers_ok && h.ok;
    }
    if (emit_rc || !helpers_ok) {
        *err = helpers_ok ? std::string("device: ") + mgx_last_error() : "short write to " + path;
        close(fd);
        return false;
    }
    stamp("stream written");
    sk.block_at.push_back(sk.at);                    // the position after the last block: virtual offsets at the very end
    ok = ok && pwrite_all(fd, kEofBlock, 28, sk.at);
    ok = (close(fd) == 0) && ok;
    if (!ok) { *err = "short write to " + path; return false; }
    auto voff = [&](size_t k, uint64_t* vb, uint64_t* ve) {
        *vb = (sk.block_at[uoff[k] / kBlockIn] << 16) | (uoff[k] % kBlockIn);
        *ve = (sk.block_at[uoff[k + 1] / kBlockIn] << 16) | (uoff[k + 1] % kBlockIn);
    };
    const bool iok = write_bai(path, hdr, recs, voff, err, threads);      // this voff holds no state: any thread, any order
    stamp("index written");
    return iok;
}

}  // namespace bamout
