// sortmardup_main.cpp -- `sortmardup`: coordinate sort + mark duplicates, SAM text in, BAM + BAI out.
//
// Same command line and outputs as the reference tool (sortmardup/main.cpp:47-78):
//     sortmardup [-I input.sam] [-t threads] -O output.bam
// text SAM from a file or stdin; output.bam is replaced if it exists; output.bam.bai is written
// next to it; stage timings go to stdout (time_stamp(), main.cpp:597-607).
//
// Ingest is a pipeline over bounded slices of the text, the shape of the reference's reader thread feeding its
// shuffle threads through a bounded queue of line blocks (main.cpp:505-562, 129-192):
//   reader (main thread)   cuts ~8 MB slices at a template boundary (a queryname group never straddles two slices,
//                          so mates are found inside their slice) and queues them; of a regular file it only reads
//                          the few KB around every cut, stdin it reads whole; the queue is bounded
//   parsers (-t threads)   pread their slice (regular file), parse it into BAM-ready records, run
//                          mgx_sortdedup_pack on it (host keys, arrival order, slice-local mate indices), hand the
//                          BAM bytes to the device record store (-z device) and build the writer's per-record view
//   commit (in slice order, by whichever parser finishes the next slice)
//                          turns mate indices into global arrival indices and hands the packed records to
//                          mgx_sortdedup_upload_chunk: staging and the PCIe copy run while later slices are
//                          still being parsed
// then mgx_sortdedup_run (MI355X: radix sorts + duplicate search) and the output: the records are gathered in sorted
// order, duplicate-flagged and BGZF-compressed on the device (-z device, mgx_bgzf_store_emit), or gathered by writer
// threads for the device compressor (-z pinned) or for zlib (-z zlib, the reference's way); BAI from the records'
// virtual offsets.
// There is no CPU fallback: without a HIP device the tool exits with an error.
#include <fcntl.h>
#include <getopt.h>
#include <malloc.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bam_writer.h"
#include "../../../include/mgx_bgzf.h"
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

struct Slice { uint64_t seq = 0; std::string text; uint64_t file_off = 0; size_t file_len = 0; bool from_file = false; };   // owned text (stdin) or a range of the input file

struct Chunk {                       // one parsed + packed slice
    std::vector<uint16_t> flag; std::vector<int32_t> tid; std::vector<int64_t> pos; std::vector<int32_t> end;
    std::vector<uint32_t> cigar; std::vector<uint64_t> cigar_off{0};
    std::vector<uint16_t> score;                                        // BAMRecord::score per record, from the parser
    std::vector<char> qname; std::vector<uint64_t> qname_off{0};
    std::vector<uint8_t> blob; std::vector<uint64_t> blob_off{0};      // BAM bytes per record (kept until the output is written)
    std::vector<mgx_rec_t> recs; std::vector<uint32_t> input_index;   // arrival order inside the slice
    uint64_t dev_base = 0;               // -z device: where the slice's BAM bytes are in HBM (blob is dropped then)
    uint64_t arrival_base = 0;           // set at commit: the slice's first arrival index
    std::vector<struct Kept> kept;       // what the writer needs per record, slice-local arrival order (filled by the parser thread)
    std::string err;
};

// what the writer needs per record, in ARRIVAL order
struct Kept { const uint8_t* blob; uint32_t len; int32_t tid, beg, end; bool mapped; };

const char* line_qname_end(const char* line, const char* end) {
    const char* t = (const char*)memchr(line, '\t', (size_t)(end - line));
    return t ? t : end;
}

// Offset at which the LAST queryname group of [data, data + size) starts (size ends on a line boundary).
size_t last_group_start(const char* data, size_t size) {
    if (size == 0) return 0;
    size_t line_end = size;                                  // one past the '\n' of the line under inspection
    auto line_begin = [&](size_t e) { size_t b = e - 1; while (b > 0 && data[b - 1] != '\n') --b; return b; };
    size_t b = line_begin(line_end);
    const char* qn = data + b; const size_t qn_len = (size_t)(line_qname_end(qn, data + line_end) - qn);
    size_t group = b;
    while (group > 0) {
        const size_t pb = line_begin(group);
        const char* pq = data + pb; const size_t pl = (size_t)(line_qname_end(pq, data + group) - pq);
        if (pl != qn_len || memcmp(pq, qn, qn_len) != 0) break;
        group = pb;
    }
    return group;
}

void parse_slice(const char* data, size_t size, const samtext::Header& h, uint64_t L_expected, Chunk* c) {
    size_t off = 0, hi = size;
    {
        // sized from the text so that the vectors do not grow by doubling (a record is rarely under 100 bytes of text)
        const size_t est = size / 100 + 16;
        c->flag.reserve(est); c->tid.reserve(est); c->pos.reserve(est); c->end.reserve(est);
        c->cigar_off.reserve(est + 1); c->score.reserve(est); c->qname_off.reserve(est + 1); c->blob_off.reserve(est + 1);
        c->cigar.reserve(est * 2); c->qname.reserve(size / 4); c->blob.reserve(size);
    }
    while (off < hi) {
        const char* nl = (const char*)memchr(data + off, '\n', hi - off);
        size_t len = nl ? (size_t)(nl - (data + off)) : hi - off;
        const size_t next = off + len + 1;
        if (len && data[off + len - 1] == '\r') --len;
        if (len) {
            samtext::Parsed pr;
            if (!samtext::parse_record_into(data + off, len, h, &pr, &c->cigar, nullptr, &c->qname, &c->blob, &c->err)) {
                c->err += " at: " + std::string(data + off, std::min<size_t>(len, 80));
                return;
            }
            c->flag.push_back(pr.flag); c->tid.push_back(pr.tid); c->pos.push_back(pr.pos); c->end.push_back(pr.end);
            c->cigar_off.push_back(c->cigar.size()); c->score.push_back(pr.score); c->qname_off.push_back(c->qname.size());
            c->blob_off.push_back(c->blob.size());
        }
        off = next;
    }
    const size_t n = c->flag.size();
    mgx_raw_records_t raw{};
    raw.n_records = n; raw.flag = c->flag.data(); raw.tid = c->tid.data(); raw.pos = c->pos.data();
    raw.cigar_off = c->cigar_off.data(); raw.cigar = c->cigar.data(); raw.qual_off = nullptr; raw.qual = nullptr;      // the score comes from the parser
    raw.qname_off = c->qname_off.data(); raw.qname = c->qname.data();
    raw.n_targets = (uint32_t)h.ref_len.size(); raw.target_len = h.ref_len.data();
    c->recs.resize(n); c->input_index.resize(n);
    uint64_t L = 0;
    if (mgx_sortdedup_pack_scored(&raw, c->score.data(), c->recs.data(), c->input_index.data(), &L)) { c->err = std::string("pack: ") + mgx_last_error(); return; }
    (void)L_expected;
    // the parse-time arrays are not needed any more (the keys are in recs); keep what the writer needs
    std::vector<uint32_t>().swap(c->cigar); std::vector<uint16_t>().swap(c->score); std::vector<char>().swap(c->qname);
    std::vector<uint64_t>().swap(c->cigar_off); std::vector<uint64_t>().swap(c->qname_off);
}

}  // namespace

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);   // the three sorts overlap on three streams: keep them on distinct hardware queues
    // The parser threads allocate and free a slice's arrays (megabytes each) thousands of times: keep them inside the malloc
    // arenas instead of one mmap / munmap -- and its page faults on fresh zero pages -- per array.
    mallopt(M_MMAP_THRESHOLD, 32 << 20);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
    const char* in_path = nullptr; const char* out_path = nullptr;
    int threads = (int)std::thread::hardware_concurrency();
    int device = 0, level = 6;
    enum { kOutDevice, kOutPinned, kOutZlib } out_mode = kOutDevice;
    size_t slice_bytes = 8u << 20;     // 8 MB: 20 M records parse in 1.2 s (32 MB slices: 2.1 s -- fewer, longer tasks per thread)
    int c;
    while ((c = getopt(argc, argv, "I:O:t:d:l:s:z:")) >= 0) {
        switch (c) {
            case 'I': in_path = optarg; break;
            case 'O': out_path = optarg; break;
            case 't': threads = atoi(optarg); break;
            case 'd': device = atoi(optarg); break;          // extension: HIP device ordinal
            case 'l': level = atoi(optarg); break;           // extension: deflate level (with -z zlib)
            case 'z':                                        // extension: where the output is made
                // device (default): BAM bytes resident in HBM from ingest on, gathered + compressed on the device
                // pinned: BAM bytes in host memory, gathered by the writer threads into pinned batches, compressed on the device
                // zlib:   BAM bytes in host memory, zlib at -l level on the writer threads (the reference's way)
                out_mode = !strcmp(optarg, "zlib") ? kOutZlib : !strcmp(optarg, "pinned") ? kOutPinned : kOutDevice;
                break;
            case 's': slice_bytes = (size_t)atoll(optarg); break;   // extension: bytes of SAM text per slice
            default: fprintf(stderr, "usage: %s [-I input.sam] [-t num] -O output.bam\n", argv[0]); return 2;
        }
    }
    if (!out_path) { fprintf(stderr, "usage: %s [-I input.sam] [-t num] -O output.bam\n", argv[0]); return 2; }
    if (threads < 1) threads = 1;
    if (slice_bytes < 1024) slice_bytes = 1024;
    g_t0 = g_last = clk::now();
    time_stamp("program start");
    unlink(out_path);                                        // main.cpp:66-68

    FILE* f = in_path ? fopen(in_path, "rb") : stdin;
    if (!f) { fprintf(stderr, "cannot read %s\n", in_path ? in_path : "stdin"); return 1; }
    uint64_t file_bytes = 0;
    { struct stat sb; if (in_path && stat(in_path, &sb) == 0) file_bytes = (uint64_t)sb.st_size; }
#ifdef F_SETPIPE_SZ
    (void)fcntl(fileno(f), F_SETPIPE_SZ, 1 << 20);          // a pipe on stdin: 1 MB instead of 64 KB per hand-over (ignored for files)
#endif
    setvbuf(f, nullptr, _IONBF, 0);                          // read_more() asks for megabytes at a time: no second buffer

    // A regular file is read by the parser threads themselves (their own slice, in place from a populated mapping -- below -- or
    // by pread into their own buffer): the reader only looks at a few KB around every cut to place it on a queryname-group
    // boundary.  (Mapping the file and letting every page fault on its own cost 1.8 M page faults per 7 GB and made the ingest time
    // vary by 50 % from run to run.)  stdin goes through read() and owned slices.
    const int in_fd = (in_path && file_bytes) ? fileno(f) : -1;
    const bool map = in_fd >= 0;
    const size_t map_size = (size_t)file_bytes;
    // Round 3: the parsers read their slice IN PLACE from a mapping of the file, after one madvise(MADV_POPULATE_READ) per slice has
    // the kernel fill in its page-table entries (2048 of them in one call, no fault per page), instead of copying the slice out of
    // the page cache with pread: 200 M records ingest 5.7 -> 4.5-5.1 s on one box.  MGX_CLI_MMAP_IN=0, a kernel without
    // MADV_POPULATE_READ or a file that cannot be mapped keep the pread path.
    const char* in_base = nullptr;
#ifdef MADV_POPULATE_READ
    {
        const char* e = getenv("MGX_CLI_MMAP_IN");
        if (map && (!e || atoi(e) != 0)) {
            void* mp = mmap(nullptr, map_size, PROT_READ, MAP_SHARED, in_fd, 0);
            if (mp != MAP_FAILED) {
                // is the advice known to this kernel? (EINVAL on kernels before 5.14)
                if (madvise(mp, std::min<size_t>(map_size, 4096), MADV_POPULATE_READ) == 0) in_base = static_cast<const char*>(mp);
                else (void)munmap(mp, map_size);
            }
        }
    }
#endif
    auto pread_all = [](int fd, char* dst, size_t n, uint64_t at) -> bool {
        while (n) {
            const ssize_t g = pread(fd, dst, n, (off_t)at);
            if (g <= 0) { if (g < 0 && errno == EINTR) continue; return false; }
            dst += g; n -= (size_t)g; at += (uint64_t)g;
        }
        return true;
    };

    // ---- header: read until a line that does not start with '@' is complete
    std::string carry;                                       // text read but not yet handed to a parser
    samtext::Header hdr;
    bool eof = false;
    std::vector<char> buf(1u << 20);
    auto read_more = [&](size_t want) {
        size_t got_total = 0;
        while (!eof && got_total < want) {
            const size_t got = fread(buf.data(), 1, std::min(buf.size(), want - got_total), f);
            if (got == 0) { eof = true; break; }
            carry.append(buf.data(), got); got_total += got;
        }
    };
    size_t map_pos = 0;
    if (map) {
        // the header: the file's head, more of it until a line that does not start with '@' is in sight
        std::vector<char> head;
        for (size_t want = 1u << 20;; want *= 4) {
            head.resize(std::min(want, map_size));
            if (!pread_all(in_fd, head.data(), head.size(), 0)) { fprintf(stderr, "cannot read %s\n", in_path); return 1; }
            size_t off = 0; bool body_seen = false;
            while (off < head.size()) {
                if (head[off] != '@') { body_seen = true; break; }
                const char* nl = (const char*)memchr(head.data() + off, '\n', head.size() - off);
                if (!nl) break;
                off = (size_t)(nl - head.data()) + 1;
            }
            if (body_seen || head.size() == map_size) break;
        }
        map_pos = samtext::parse_header(head.data(), head.size(), &hdr);
    } else for (;;) {
        // the header is complete once the buffer holds a full line that does not start with '@'
        size_t off = 0; bool body_seen = false;
        while (off < carry.size()) {
            if (carry[off] != '@') { body_seen = true; break; }
            const size_t nl = carry.find('\n', off);
            if (nl == std::string::npos) break;
            off = nl + 1;
        }
        if (body_seen || eof) break;
        read_more(1u << 20);
    }
    if (!map) {
        const size_t body = samtext::parse_header(carry.data(), carry.size(), &hdr);
        carry.erase(0, body);
    }
    uint64_t L = 0;
    for (uint64_t x : hdr.ref_len) L += x;

    // The device contexts come up on a thread of their own (the HIP runtime takes a few tenths of a second to start)
    // while the first slices are already being parsed; a parser waits for them only when it has bytes for the device.
    mgx_bgzf_t* zctx = nullptr; mgx_bgzf_store_t* store = nullptr;
    mgx_sortdedup_t* sd = nullptr;
    bool use_store = out_mode == kOutDevice;
    std::mutex gpu_mu; std::condition_variable gpu_cv; int gpu_state = 0;          // 0 starting, 1 ready, -1 failed
    std::string gpu_error;
    std::thread gpu_init([&]() {
        bool ok = true;
        const bool tr = getenv("MGX_CLI_TRACE") != nullptr;
        const auto i0 = clk::now();
        auto since = [&]() { return std::chrono::duration<double>(clk::now() - i0).count(); };
        // What the parsers wait for comes first: the record store (runtime start, streams, the first piece of HBM) and the
        // sort context; the compressor's own state is set up afterwards, while the text is being parsed.
        if (use_store && file_bytes) {
            // -z device keeps every BAM byte in HBM next to the sort's buffers; when the input cannot fit, say so now and keep
            // the bytes in host memory instead of failing in the middle of the ingest.  BAM bytes are about half
            // the SAM text; the sort and the emit need about 100 bytes per record of ~360 text bytes on top.
            uint64_t free_b = 0, total_b = 0;
            if (mgx_bgzf_device_memory(device, &free_b, &total_b) == 0) {
                if (const char* e = getenv("MGX_CLI_DEVICE_FREE")) free_b = strtoull(e, nullptr, 10);      // tests: pretend
                const uint64_t need = file_bytes * 6 / 10 + (file_bytes / 360) * 100 + (1ull << 30);
                if (need > free_b) {
                    fprintf(stderr, "sortmardup: about %.1f GB of device memory needed for -z device, %.1f GB free: keeping the BAM bytes in host memory (-z pinned)\n",
                            need / 1e9, free_b / 1e9);
                    use_store = false; out_mode = kOutPinned;
                }
            }
        }
        if (tr) fprintf(stderr, "  bring-up: runtime up, device memory known at %.3f s\n", since());
        if (use_store && mgx_bgzf_create(device, 0, &zctx)) ok = false;
        if (tr) fprintf(stderr, "  bring-up: compressor context at %.3f s\n", since());
        if (ok && use_store && (mgx_bgzf_store_create(zctx, &store) || mgx_bgzf_store_reserve(store, file_bytes ? file_bytes * 6 / 10 : (1ull << 30)))) ok = false;
        if (tr) fprintf(stderr, "  bring-up: record store ready at %.3f s\n", since());
        if (ok && (mgx_sortdedup_create(device, 0, &sd) || mgx_sortdedup_upload_begin(sd, L, file_bytes / 256))) ok = false;
        if (tr) fprintf(stderr, "  bring-up: sort context ready at %.3f s\n", since());
        {
            std::lock_guard<std::mutex> g(gpu_mu);
            if (!ok) gpu_error = mgx_last_error();
            gpu_state = ok ? 1 : -1;
            gpu_cv.notify_all();
        }
        if (ok && zctx && mgx_bgzf_prepare(zctx)) fprintf(stderr, "sortmardup: %s\n", mgx_last_error());      // not fatal here: the output stage reports it
        if (tr) fprintf(stderr, "  bring-up: compressor ready at %.3f s\n", since());
    });
    auto gpu_ready = [&]() -> bool {
        std::unique_lock<std::mutex> lk(gpu_mu);
        gpu_cv.wait(lk, [&] { return gpu_state != 0; });
        return gpu_state == 1;
    };

    // ---- the pipeline
    std::mutex mu;
    std::condition_variable cv_work, cv_room;
    std::deque<Slice> queue;
    const size_t queue_cap = (size_t)threads * 2;
    bool done_reading = false;
    std::atomic<bool> failed{false};
    std::string first_error;
    // commit state (guarded by commit_mu)
    std::mutex commit_mu;
    std::map<uint64_t, std::unique_ptr<Chunk>> ready;
    uint64_t next_commit = 0, n_total = 0;
    std::vector<std::unique_ptr<Chunk>> kept_chunks;         // committed slices: their writer records, and (not -z device) the BAM bytes
    bamout::NoInitVector<Kept> by_arrival;                   // flattened after the ingest (its pages first touched by the copying gang)
    auto fail = [&](const std::string& msg) {
        std::lock_guard<std::mutex> g(mu);
        if (!failed.exchange(true)) first_error = msg;
        cv_work.notify_all(); cv_room.notify_all();
    };
    double commit_seconds = 0, upload_seconds = 0;           // serial part of the ingest (MGX_CLI_TRACE)
    // The in-order commit is the one serial step of the ingest: it only turns slice-local mate indices into arrival
    // indices and hands the packed records to the upload; what it frees is handed back to die outside the lock.
    auto commit_ready = [&](std::vector<std::vector<mgx_rec_t>>* trash) {      // called with commit_mu held
        const auto c0 = clk::now();
        struct Acc { double* d; clk::time_point t; ~Acc() { *d += std::chrono::duration<double>(clk::now() - t).count(); } } acc{&commit_seconds, c0};
        for (;;) {
            auto it = ready.find(next_commit);
            if (it == ready.end()) return;
            std::unique_ptr<Chunk> ch = std::move(it->second);
            ready.erase(it);
            const uint64_t base = n_total, n = ch->recs.size();
            if (base + n >= 0xFFFFFFF0ull) { fail("more than 2^32 records"); return; }
            for (auto& r : ch->recs) if (r.mate != MGX_NO_MATE) r.mate += (uint32_t)base;        // slice-local -> global arrival index
            const auto u0 = clk::now();
            if (n && mgx_sortdedup_upload_chunk(sd, base, n, ch->recs.data())) { fail(std::string("GPU: ") + mgx_last_error()); return; }
            upload_seconds += std::chrono::duration<double>(clk::now() - u0).count();
            ch->arrival_base = base;
            n_total += n;
            trash->emplace_back(std::move(ch->recs));
            kept_chunks.push_back(std::move(ch));
            ++next_commit;
        }
    };
    auto gpu_state_now = [&]() -> int { std::lock_guard<std::mutex> lk(gpu_mu); return gpu_state; };
    // what a parsed slice still needs of the device: its BAM bytes into the record store, the writer's view of its records, the
    // in-order commit (upload of the packed records)
    auto finish_chunk = [&](std::unique_ptr<Chunk> ch, uint64_t seq) -> bool {
        if (store) {                                    // the slice's BAM bytes go to HBM now and leave host memory
            if (mgx_bgzf_store_put(store, ch->blob.data(), ch->blob.size(), &ch->dev_base)) { fail(std::string("GPU: ") + mgx_last_error()); return false; }
            std::vector<uint8_t>().swap(ch->blob);
        }
        {
            // the writer's view of every record, in the slice's arrival order; the parse-time arrays die here
            const size_t n = ch->recs.size();
            ch->kept.resize(n);
            for (size_t k = 0; k < n; ++k) {
                const uint32_t src = ch->input_index[k];
                const uint8_t* where = store ? (const uint8_t*)(uintptr_t)(ch->dev_base + ch->blob_off[src]) : ch->blob.data() + ch->blob_off[src];
                ch->kept[k] = Kept{where, (uint32_t)(ch->blob_off[src + 1] - ch->blob_off[src]), ch->tid[src], (int32_t)ch->pos[src], ch->end[src],
                                   (ch->flag[src] & 4) == 0};
            }
            std::vector<uint32_t>().swap(ch->input_index);
            std::vector<uint16_t>().swap(ch->flag); std::vector<int32_t>().swap(ch->tid); std::vector<int64_t>().swap(ch->pos);
            std::vector<int32_t>().swap(ch->end); std::vector<uint64_t>().swap(ch->blob_off);
        }
        std::vector<std::vector<mgx_rec_t>> trash;
        {
            std::lock_guard<std::mutex> g(commit_mu);
            ready.emplace(seq, std::move(ch));
            commit_ready(&trash);
        }
        return !failed.load();
    };
    auto worker = [&]() {
        std::vector<char> text_buf;                          // this thread's slice of the input file
        // The HIP runtime and the contexts take 0.2-0.3 s to come up: until then a parser keeps its parsed slices and goes on with
        // the next one instead of waiting with one slice in hand (round 3: at 4 M records the parse itself is 0.05 s of thread
        // time per thread -- the ingest was the bring-up plus everything that waited for it)
        std::vector<std::pair<uint64_t, std::unique_ptr<Chunk>>> waiting;
        for (;;) {
            Slice sl;
            bool have = false;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return !queue.empty() || done_reading || failed.load(); });
                if (failed.load()) return;
                if (!queue.empty()) {
                    sl = std::move(queue.front());
                    queue.pop_front();
                    cv_room.notify_one();
                    have = true;
                }
            }
            std::unique_ptr<Chunk> ch;
            if (have) {
                ch.reset(new Chunk);
                if (sl.from_file && in_base) {
#ifdef MADV_POPULATE_READ
                    const uintptr_t a0 = (uintptr_t)(in_base + sl.file_off) & ~(uintptr_t)4095, a1 = ((uintptr_t)(in_base + sl.file_off + sl.file_len) + 4095) & ~(uintptr_t)4095;
                    (void)madvise((void*)a0, a1 - a0, MADV_POPULATE_READ);
                    parse_slice(in_base + sl.file_off, sl.file_len, hdr, L, ch.get());
                    (void)madvise((void*)a0, a1 - a0, MADV_DONTNEED);      // drops this process's entries only: the pages stay in the page cache
#endif
                } else if (sl.from_file) {
                    if (text_buf.size() < sl.file_len) text_buf.resize(sl.file_len);
                    if (!pread_all(in_fd, text_buf.data(), sl.file_len, sl.file_off)) { fail("read error on the input file"); return; }
                    parse_slice(text_buf.data(), sl.file_len, hdr, L, ch.get());
                } else parse_slice(sl.text.data(), sl.text.size(), hdr, L, ch.get());
                std::string().swap(sl.text);
                if (!ch->err.empty()) { fail("SAM parse error: " + ch->err); return; }
                // (bounded: 12 slices per thread -- ~80 MB of parsed records, 1.3 GB over 16 threads -- cover a 4 M-record input whole;
                // with 32 the peak resident set of a 20 M-record run grew from 5.8 to 9.2 GB for 0.1 s of ingest)
                if (waiting.size() < 12 && gpu_state_now() == 0) { waiting.emplace_back(sl.seq, std::move(ch)); continue; }
            }
            if (!have && waiting.empty()) return;           // the input is used up and nothing of this thread's waits
            if (!gpu_ready()) { fail("GPU: " + gpu_error); return; }
            for (auto& w : waiting) if (!finish_chunk(std::move(w.second), w.first)) return;
            waiting.clear();
            if (have) { if (!finish_chunk(std::move(ch), sl.seq)) return; }
            else return;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(worker);

    // ---- reader: slices end where a queryname group ends
    uint64_t seq = 0;
    size_t window = slice_bytes;                             // grows only while one queryname group fills the whole window
    std::vector<char> tail;
    size_t tail_want = 64u << 10;                            // how much text before a tentative cut is inspected
    while (map && !failed.load() && map_pos < map_size) {
        const size_t limit = std::min(map_size - map_pos, window);
        size_t cut;
        if (limit == map_size - map_pos) cut = limit;
        else {
            // the last complete line before map_pos + limit, and where its queryname group starts
            const size_t tw = std::min(limit, tail_want);
            const uint64_t a0 = map_pos + limit - tw;
            tail.resize(tw);
            if (!pread_all(in_fd, tail.data(), tw, a0)) { fail("read error on the input file"); break; }
            const char* last_nl = (const char*)memrchr(tail.data(), '\n', tw);
            size_t first = 0;                                // first byte of the first COMPLETE line in the tail
            if (a0 > map_pos) { const char* nl0 = (const char*)memchr(tail.data(), '\n', tw); first = nl0 ? (size_t)(nl0 - tail.data()) + 1 : tw; }
            size_t g = 0;
            const bool have = last_nl && (size_t)(last_nl - tail.data()) + 1 > first;
            if (have) g = first + last_group_start(tail.data() + first, (size_t)(last_nl - tail.data()) + 1 - first);
            if (!have || g == first) {
                // the group reaches the head of what was inspected: look further back, or (the whole window is one
                // group) further ahead
                if (a0 > map_pos) { tail_want *= 4; continue; }
                if (!have || g == 0) { window *= 2; tail_want = 64u << 10; continue; }
            }
            cut = (size_t)(a0 - map_pos) + g;
            if (cut == 0) { window *= 2; continue; }
        }
        window = slice_bytes; tail_want = 64u << 10;
        Slice sl;
        sl.seq = seq++;
        sl.from_file = true; sl.file_off = map_pos; sl.file_len = cut;
        map_pos += cut;
        std::unique_lock<std::mutex> lk(mu);
        cv_room.wait(lk, [&] { return queue.size() < queue_cap || failed.load(); });
        if (failed.load()) break;
        queue.push_back(std::move(sl));
        cv_work.notify_one();
    }
    while (!map && !failed.load()) {
        if (carry.size() < window && !eof) read_more(window - carry.size());
        if (carry.empty() && eof) break;
        size_t cut;
        if (eof && carry.size() <= window) cut = carry.size();
        else {
            const size_t limit = std::min(carry.size(), window);
            const size_t last_nl = carry.rfind('\n', limit - 1);
            cut = last_nl == std::string::npos ? 0 : last_group_start(carry.data(), last_nl + 1);
            if (cut == 0) { window *= 2; continue; }         // not one complete group in the window yet: look at more text
        }
        window = slice_bytes;
        Slice sl;
        sl.seq = seq++;
        sl.text.assign(carry.data(), cut);
        carry.erase(0, cut);
        std::unique_lock<std::mutex> lk(mu);
        cv_room.wait(lk, [&] { return queue.size() < queue_cap || failed.load(); });
        if (failed.load()) break;
        queue.push_back(std::move(sl));
        cv_work.notify_one();
    }
    {
        std::lock_guard<std::mutex> g(mu);
        done_reading = true;
        cv_work.notify_all();
    }
    for (auto& th : pool) th.join();
    gpu_init.join();
    if (!gpu_ready()) { fprintf(stderr, "GPU: %s\n", gpu_error.c_str()); return 1; }
    if (in_path) fclose(f);
    if (failed.load()) { fprintf(stderr, "%s\n", first_error.c_str()); return 1; }
    const size_t n = (size_t)n_total;
    {
        by_arrival.resize(n);
        std::vector<std::thread> gang;
        std::atomic<size_t> next_chunk{0};
        for (int t = 0; t < std::max(1, std::min(threads, 16)); ++t)
            gang.emplace_back([&]() {
                for (size_t i; (i = next_chunk.fetch_add(1)) < kept_chunks.size();) {
                    Chunk& ch = *kept_chunks[i];
                    if (!ch.kept.empty()) memcpy(&by_arrival[ch.arrival_base], ch.kept.data(), ch.kept.size() * sizeof(Kept));
                    std::vector<Kept>().swap(ch.kept);
                }
            });
        for (auto& th : gang) th.join();
    }
    printf("%zu alignment records, %zu reference sequences, %llu slices\n", n, hdr.ref_name.size(), (unsigned long long)seq);
    time_stamp("read + parse + pair + upload done");
    if (getenv("MGX_CLI_TRACE")) fprintf(stderr, "  ingest: %.3f s inside the in-order commit (one thread at a time), %.3f s of it in mgx_sortdedup_upload_chunk\n", commit_seconds, upload_seconds);

    if (mgx_sortdedup_upload_end(sd, n)) { fprintf(stderr, "GPU: %s\n", mgx_last_error()); return 1; }
    bamout::NoInitVector<uint32_t> order(n); bamout::NoInitVector<uint8_t> dup(n);
    {
        // the results land in fresh memory: its pages are touched by all threads first (a device-to-host copy into untouched
        // pageable memory faults them in one by one on the runtime's copy path: up to 0.5 s for the 1 GB of 200 M records)
        std::vector<std::thread> gang;
        const size_t T = (size_t)std::max(1, std::min(threads, 16));
        for (size_t t = 0; t < T; ++t)
            gang.emplace_back([&, t]() {
                auto touch = [&](uint8_t* p, size_t bytes) { for (size_t o = bytes * t / T & ~(size_t)4095, e = bytes * (t + 1) / T; o < e; o += 4096) p[o] = 0; };
                touch(reinterpret_cast<uint8_t*>(order.data()), n * sizeof(uint32_t));
                touch(dup.data(), n);
            });
        for (auto& th : gang) th.join();
    }
    if (mgx_sortdedup_run(sd) || mgx_sortdedup_results(sd, order.data(), dup.data())) { fprintf(stderr, "GPU: %s\n", mgx_last_error()); return 1; }
    mgx_sortdedup_stats_t st{};
    mgx_sortdedup_stats(sd, &st);
    printf("double pairs %llu, single pairs %llu, records marked duplicate %llu, device pipeline %.3f ms\n",
           (unsigned long long)st.n_double, (unsigned long long)st.n_single, (unsigned long long)st.n_dup_records, st.ms_total);
    mgx_sortdedup_destroy(sd);
    time_stamp("sort + duplicate search done");

    // ---- mark + compress + write
    bamout::RecordRefs out(n);
    {
        // a gather with random reads from by_arrival: spread over the threads
        std::vector<std::thread> gang;
        const size_t T = (size_t)std::max(1, std::min(threads, 16));
        for (size_t t = 0; t < T; ++t)
            gang.emplace_back([&, t]() {
                for (size_t q = n * t / T, e = n * (t + 1) / T; q < e; ++q) {
                    const uint32_t arrival = order[q];
                    const Kept& k = by_arrival[arrival];
                    out[q] = bamout::RecordRef{k.blob, k.len, k.tid, k.beg, k.end, dup[arrival] != 0, k.mapped};
                }
            });
        for (auto& th : gang) th.join();
    }
    // what the output stage no longer needs goes back to the kernel on a thread of its own while the stream is written (at 200 M
    // records: 7 GB of per-record bookkeeping that would otherwise be torn down after the last byte is on disk)
    // (madvise, not free: unmapping takes the address space's lock for writing for as long as it frees pages, and the output
    // stage's threads are first-touching their own arrays right now; dropping the pages only needs it for reading)
    std::atomic<bool> writer_returned{false};
    std::thread reaper([&]() {
        while (store && !bamout::g_store_arrays_ready.load() && !writer_returned.load()) std::this_thread::sleep_for(std::chrono::milliseconds(2));
        auto drop = [](void* p, size_t bytes) {
            const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
            if (e > a) (void)madvise((void*)a, e - a, MADV_DONTNEED);
        };
        drop(by_arrival.data(), by_arrival.size() * sizeof(Kept));
        drop(order.data(), order.size() * sizeof(uint32_t));
        drop(dup.data(), dup.size());
    });
    std::string err;
    const bool wrote = store ? bamout::write_bam_store(out_path, hdr, out, zctx, store, threads, &err)
                             : bamout::write_bam(out_path, hdr, out, threads, level, out_mode == kOutPinned ? device : -1, &err);
    writer_returned.store(true);
    reaper.join();
    if (!wrote) { fprintf(stderr, "write: %s\n", err.c_str()); return 1; }
    time_stamp("output done");
    // Both files are closed.  What is left is tearing down ~N small records' bookkeeping, the arenas and the HIP runtime --
    // a few tenths of a second at 20 M records that change nothing on disk: leave it to the kernel.
    fflush(stdout); fflush(stderr);
    _exit(0);
}
