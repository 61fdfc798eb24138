// tests/cpp/test_host_sanitize.cpp -- the host-only pieces of the library (record packing, the multi-GPU router and
// merge, the PairHMM batch packer) driven over seeded random inputs; built by tests/test_host_sanitizers.py with
// -fsanitize=address,undefined and with -fsanitize=thread (no HIP, no device).  Exit code 0 = nothing reported.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mgx_pairhmm.h"
#include "mgx_sortdedup.h"
#include "../../fast-genomic-data-processing_amd/csrc/pairhmm_pack.h"
#include "../../fast-genomic-data-processing_amd/csrc/cli/sam_text.h"

static uint64_t rng_state = 0x5EED;
static uint64_t rnd() { rng_state += 0x9E3779B97F4A7C15ull; uint64_t z = rng_state; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

static int route_case(uint64_t n_templates, uint32_t n_shards, uint64_t L) {
    std::vector<mgx_rec_t> recs;
    for (uint64_t t = 0; t < n_templates; ++t) {
        const uint64_t kind = rnd() % 10;
        mgx_rec_t a{}, b{};
        a.coord = rnd() % (L + 1); a.prime5 = a.coord + rnd() % 5; a.flag = (uint16_t)(1 | (rnd() & 16)); a.score = (uint16_t)rnd(); a.tile = (uint16_t)t;
        if (kind == 0) { a.mate = MGX_NO_MATE; a.flag |= 8; recs.push_back(a); continue; }             // fragment
        if (kind == 1) { a.mate = MGX_NO_MATE; a.flag |= 4; recs.push_back(a); continue; }             // unmapped: ordering only
        b = a; b.coord = rnd() % (L + 1); b.prime5 = kind == 2 ? ~0ull - rnd() % 7 : b.coord + rnd() % 900;  // sometimes a wrapped 5' end
        b.flag = (uint16_t)(1 | 128 | (rnd() & 16));
        const uint32_t i = (uint32_t)recs.size();
        a.mate = i + 1; b.mate = i;
        recs.push_back(a); recs.push_back(b);
    }
    mgx_sortdedup_routed_t* r = nullptr;
    if (mgx_sortdedup_route(L, recs.size(), recs.data(), n_shards, -1, &r)) return 1;
    uint64_t n_order = 0, n_mark = 0;
    std::vector<uint32_t> order(recs.size(), 0); std::vector<uint8_t> dup(recs.size(), 0);
    for (uint32_t k = 0; k < n_shards; ++k) {
        mgx_sortdedup_shard_t sh;
        if (mgx_sortdedup_routed_shard(r, k, &sh)) return 2;
        n_order += sh.n_order; n_mark += sh.n_mark;
        for (uint64_t i = 0; i < sh.n_mark; ++i) if (sh.mark_recs[i].mate != MGX_NO_MATE && sh.mark_recs[i].mate >= sh.n_mark) return 3;
        std::vector<uint32_t> so(sh.order_arrival, sh.order_arrival + sh.n_order);
        std::vector<uint8_t> sd(sh.n_mark, 1);
        if (mgx_sortdedup_merge(r, k, so.data(), sd.data(), order.data(), dup.data())) return 4;
    }
    uint64_t n_nonign = 0;
    for (auto& x : recs) if (!(x.flag & (4 | 256 | 2048))) ++n_nonign;
    mgx_sortdedup_routed_free(r);
    return (n_order == recs.size() && n_mark == n_nonign) ? 0 : 5;
}

static int pack_case(uint64_t n_reads, uint64_t n_haps, uint64_t n_pairs, bool cross) {
    std::vector<uint64_t> roff(1, 0), hoff(1, 0);
    for (uint64_t r = 0; r < n_reads; ++r) roff.push_back(roff.back() + 1 + rnd() % 40);
    for (uint64_t h = 0; h < n_haps; ++h) hoff.push_back(hoff.back() + 1 + rnd() % 60);
    std::vector<uint8_t> rb(roff.back(), 'A'), hb(hoff.back(), 'C');
    std::vector<uint32_t> pr, ph;
    for (uint64_t i = 0; i < n_pairs; ++i) { pr.push_back((uint32_t)(rnd() % n_reads)); ph.push_back((uint32_t)(rnd() % n_haps)); }
    mgx_pairhmm_input_t in{};
    in.n_reads = n_reads; in.read_off = roff.data(); in.bases = in.qual = in.ins = in.del = in.gcp = rb.data();
    in.n_haps = n_haps; in.hap_off = hoff.data(); in.hap_bases = hb.data();
    in.n_pairs = cross ? n_reads * n_haps : n_pairs; in.pair_read = cross ? nullptr : pr.data(); in.pair_hap = cross ? nullptr : ph.data();
    const uint64_t total = cross ? n_reads * n_haps : n_pairs;
    for (uint64_t lo = 0; lo < total; lo += 97) {
        const uint64_t hi = lo + 97 < total ? lo + 97 : total;
        size_t need = 0;
        mgx_pairhmm_input_t out{};
        if (mgx_pairhmm_pack_batch(&in, lo, hi, nullptr, 0, &out, &need) != -28) return 10;
        std::vector<uint8_t> buf(need);
        if (mgx_pairhmm_pack_batch(&in, lo, hi, buf.data(), buf.size(), &out, &need)) return 11;
        if (out.n_pairs != hi - lo) return 12;
        for (uint64_t i = 0; i < out.n_pairs; ++i) if (out.pair_read[i] >= out.n_reads || out.pair_hap[i] >= out.n_haps) return 13;
    }
    return 0;
}

// The CLI's SAM text parser on well-formed, odd and broken lines (its integer parser and 4-bit sequence packing are
// hand-written): every line either parses to the expected fields or is rejected with a message, nothing is read out of bounds.
static int sam_case() {
    samtext::Header h;
    const std::string head = "@HD\tVN:1.6\n@SQ\tSN:chr1\tLN:100000\n@SQ\tSN:chr2\tLN:5000\n";
    if (samtext::parse_header(head.data(), head.size(), &h) != head.size() || h.ref_name.size() != 2 || h.ref_len[1] != 5000) return 10;
    struct Case { const char* line; bool ok; int32_t pos; uint32_t l_seq; size_t n_cigar; size_t aux_bytes; };
    const Case cases[] = {
        {"r1\t99\tchr1\t1001\t60\t5M2I3M\t=\t1200\t300\tACGTNACGTA\tIIIIIIIIII\tNM:i:3\tRG:Z:g\tXA:A:c\tXF:f:1.5\tXB:B:s,-1,2,300", true, 1000, 10, 3, 4 + 5 + 4 + 7 + 14},        // NM:C, RG:Z, XA:A, XF:f, XB:B:s x 3
        {"r2\t4\t*\t0\t0\t*\t*\t0\t0\tACG\t*", true, -1, 3, 0, 0},                    // odd length, no qualities, unmapped
        {"r3\t16\tchr2\t+17\t0\t3S1M\tchr1\t-0\t-2147483647\tacgt\t!!!!\tXI:i:-40000\tXJ:i:4000000000", true, 16, 4, 2, 3 + 4 + 3 + 4},
        {"r4\t0\tchr1\t1\t0\t*\t*\t0\t0\t*\t*", true, 0, 0, 0, 0},
        {"r5\t0\tchr9\t1\t0\t*\t*\t0\t0\t*\t*", false, 0, 0, 0, 0},                      // unknown reference
        {"r6\t0x10\tchr1\t1\t0\t*\t*\t0\t0\t*\t*", false, 0, 0, 0, 0},                   // FLAG is not decimal
        {"r7\t0\tchr1\t99999999999999999999\t0\t*\t*\t0\t0\t*\t*", false, 0, 0, 0, 0},   // POS overflows
        {"r8\t0\tchr1\t1\t0\t4M\t*\t0\t0\tACGT\tII", false, 0, 0, 0, 0},                 // SEQ and QUAL differ in length
        {"r9\t0\tchr1\t1\t0\tM4\t*\t0\t0\t*\t*", false, 0, 0, 0, 0},                     // CIGAR without a count
        {"r10\t0\tchr1\t1", false, 0, 0, 0, 0},                                              // fewer than 11 fields
        {"r11\t0\tchr1\t1\t0\t*\t*\t0\t0\t*\t*\tNM:i:", false, 0, 0, 0, 0},             // empty integer tag
        {"r12\t0\tchr1\t-\t0\t*\t*\t0\t0\t*\t*", false, 0, 0, 0, 0},                     // a sign alone
    };
    samtext::Record r; std::string err;
    for (const Case& c : cases) {
        const std::string line(c.line);                       // exact-size heap copy: an over-read is an ASan report
        std::vector<char> exact(line.begin(), line.end());
        err.clear();
        const bool ok = samtext::parse_record(exact.data(), exact.size(), h, &r, &err);
        if (ok != c.ok) { fprintf(stderr, "sam_case: '%s' parsed=%d (%s)\n", c.line, (int)ok, err.c_str()); return 11; }
        {
            // the one-pass parser the CLI uses: the same verdict, and the bytes encode_record makes of the Record
            std::vector<uint32_t> cg{7u}; std::vector<uint8_t> ql{9}; std::vector<char> qn{'x'}; std::vector<uint8_t> blob{1, 2, 3}, want{1, 2, 3};
            samtext::Parsed pr{}; std::string err2;
            const bool ok2 = samtext::parse_record_into(exact.data(), exact.size(), h, &pr, &cg, &ql, &qn, &blob, &err2);
            if (ok2 != ok) { fprintf(stderr, "sam_case: one-pass parser disagrees on '%s' (%s)\n", c.line, err2.c_str()); return 18; }
            if (!ok2) { if (cg.size() != 1 || ql.size() != 1 || qn.size() != 1 || blob.size() != 3 || err2.empty()) return 19; }
            else {
                bamout::encode_record(r, &want);
                if (blob != want || pr.flag != r.flag || pr.tid != r.tid || pr.pos != r.pos || pr.end != r.end()) { fprintf(stderr, "sam_case: one-pass bytes differ on '%s'\n", c.line); return 20; }
                if (cg.size() != 1 + r.cigar.size() || memcmp(cg.data() + 1, r.cigar.data(), 4 * r.cigar.size()) != 0) return 21;
                if (ql.size() != 1 + r.qual.size() || memcmp(ql.data() + 1, r.qual.data(), r.qual.size()) != 0) return 22;
                if (qn.size() != 1 + r.qname.size() || memcmp(qn.data() + 1, r.qname.data(), r.qname.size()) != 0) return 23;
            }
        }
        if (!ok) { if (err.empty()) return 12; continue; }
        if (r.pos != c.pos || r.l_seq != c.l_seq || r.cigar.size() != c.n_cigar || r.seq4.size() != (c.l_seq + 1) / 2 || r.qual.size() != c.l_seq) return 13;
        if (c.aux_bytes && r.aux.size() != c.aux_bytes) { fprintf(stderr, "sam_case: aux of '%s' is %zu bytes\n", c.line, r.aux.size()); return 14; }
    }
    // mutated copies of the valid lines (a byte replaced, dropped or doubled): the two parsers agree on the verdict and the bytes
    for (int it = 0; it < 20000; ++it) {
        std::string line(cases[rnd() % 4].line);
        for (int m = 0, nm = 1 + (int)(rnd() % 3); m < nm && !line.empty(); ++m) {
            const size_t at = rnd() % line.size();
            const char repl[] = "\t*:=0123456789-+ACGTNMIDSHXBZifsc,.!~";
            switch (rnd() % 3) {
                case 0: line[at] = repl[rnd() % (sizeof repl - 1)]; break;
                case 1: line.erase(at, 1); break;
                default: line.insert(at, 1, line[at]); break;
            }
        }
        std::vector<char> exact(line.begin(), line.end());
        std::string e1, e2;
        samtext::Record r1;
        std::vector<uint32_t> cg; std::vector<uint8_t> ql; std::vector<char> qn; std::vector<uint8_t> blob, want;
        samtext::Parsed pr{};
        const bool o1 = samtext::parse_record(exact.data(), exact.size(), h, &r1, &e1);
        const bool o2 = samtext::parse_record_into(exact.data(), exact.size(), h, &pr, &cg, &ql, &qn, &blob, &e2);
        if (o1 != o2) { fprintf(stderr, "sam_case: parsers disagree on '%s' (%s | %s)\n", line.c_str(), e1.c_str(), e2.c_str()); return 24; }
        if (o1) {
            bamout::encode_record(r1, &want);
            if (blob != want) { fprintf(stderr, "sam_case: bytes differ on '%s'\n", line.c_str()); return 25; }
            // the score the parser hands to mgx_sortdedup_pack_scored is BAMRecord::score of the record's qualities, and the
            // parse without a separate quality array (what the CLI asks for) makes the same bytes
            uint16_t sc = 0;
            for (uint8_t q : r1.qual) sc = (uint16_t)(sc + (q >= 15 ? q : 0));
            if (pr.score != sc || ql != r1.qual) { fprintf(stderr, "sam_case: score %u != %u on '%s'\n", pr.score, sc, line.c_str()); return 27; }
            std::vector<uint32_t> cg2; std::vector<char> qn2; std::vector<uint8_t> blob2; samtext::Parsed pr2{}; std::string e3;
            if (!samtext::parse_record_into(exact.data(), exact.size(), h, &pr2, &cg2, nullptr, &qn2, &blob2, &e3) || blob2 != want || pr2.score != sc) return 28;
        }
        else if (!blob.empty() || !cg.empty() || !ql.empty() || !qn.empty()) return 26;
    }
    // reads of 0 .. 200 bases over the alphabets the vector paths of the one-pass parser tell apart (sixteen A C G T N at a time,
    // either case; anything else -- IUPAC codes, '=', '.', junk -- through the byte table), qualities over the whole printable range
    // and below it: bytes and BAMRecord::score equal to the two-step parser's, on exact-size heap copies
    for (int it = 0; it < 6000; ++it) {
        const size_t l = rnd() % 201;
        const char* alpha[] = {"ACGT", "ACGTN", "ACGTNacgtn", "ACGTNacgtnMRWSYKVHDB=.", "ACGT\x01~ -"};
        const char* al = alpha[rnd() % 5];
        const size_t na = strlen(al);
        std::string seq, qual;
        for (size_t i = 0; i < l; ++i) { seq.push_back(rnd() % 50 ? al[rnd() % na] : alpha[3][rnd() % 22]); qual.push_back((char)(rnd() % 17 ? 33 + rnd() % 94 : 1 + rnd() % 32)); }
        for (char& ch : qual) if (ch == '\t' || ch == '\n') ch = '!';
        if (l == 0) { seq = "*"; qual = "*"; }
        else if (rnd() % 9 == 0) qual = "*";
        const std::string cig = l ? std::to_string(l) + "M" : std::string("*");
        const std::string line = "q" + std::to_string(it) + "\t" + std::to_string(rnd() % 4096) + "\tchr1\t" + std::to_string(1 + rnd() % 900) + "\t60\t" + cig + "\t=\t7\t0\t" + seq + "\t" + qual +
                                 (rnd() % 2 ? "\tNM:i:3\tRG:Z:grp" : "");
        std::vector<char> exact(line.begin(), line.end());
        std::string e1, e2;
        samtext::Record r1; samtext::Parsed pr{};
        std::vector<uint32_t> cg; std::vector<char> qn; std::vector<uint8_t> blob, want;
        const bool o1 = samtext::parse_record(exact.data(), exact.size(), h, &r1, &e1);
        const bool o2 = samtext::parse_record_into(exact.data(), exact.size(), h, &pr, &cg, nullptr, &qn, &blob, &e2);
        if (o1 != o2) { fprintf(stderr, "sam_case: parsers disagree on a random read of %zu bases (%s | %s)\n", l, e1.c_str(), e2.c_str()); return 29; }
        if (!o1) continue;
        bamout::encode_record(r1, &want);
        uint16_t sc = 0;
        for (uint8_t q : r1.qual) sc = (uint16_t)(sc + (q >= 15 ? q : 0));
        if (blob != want || pr.score != sc) { fprintf(stderr, "sam_case: random read of %zu bases: bytes or score differ\n", l); return 30; }
    }
    // r1's packed bases: A C G T N A C G T A -> 1 2 4 8 15 1 2 4 8 1
    err.clear();
    const std::string l1(cases[0].line);
    if (!samtext::parse_record(l1.data(), l1.size(), h, &r, &err)) return 15;
    const uint8_t want[5] = {0x12, 0x48, 0xF1, 0x24, 0x81};
    if (memcmp(r.seq4.data(), want, 5) != 0 || r.qual[0] != 'I' - 33 || r.mtid != 0 || r.mpos != 1199 || r.tlen != 300) return 16;
    const std::string l3(cases[2].line);
    if (!samtext::parse_record(l3.data(), l3.size(), h, &r, &err) || r.tlen != -2147483647 || r.mpos != -1 || r.tid != 1 || r.mtid != 0) return 17;
    return 0;
}

int main() {
    int rc = sam_case();
    for (int it = 0; it < 40 && !rc; ++it) rc = route_case(1 + rnd() % 30000, 1 + (uint32_t)(rnd() % 9), 1000 + rnd() % 5000000);
    if (!rc) rc = route_case(400000, 8, 3100000000ull);         // large enough for the router's threads to split the work
    for (int it = 0; it < 20 && !rc; ++it) rc = pack_case(1 + rnd() % 200, 1 + rnd() % 50, rnd() % 3000, it % 3 == 0);
    printf("host sanitize driver: rc %d\n", rc);
    return rc;
}
