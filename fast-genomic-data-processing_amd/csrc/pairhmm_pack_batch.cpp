// pairhmm_pack_batch.cpp -- mgx_pairhmm_pack_batch (include/mgx_pairhmm.h): the work queue's packer as a host-only
// entry point.  No HIP in this translation unit: it is also what tests/test_host_sanitizers.py builds with
// -fsanitize=address,undefined / thread.
#include <cstring>

#include "pairhmm_pack.h"

using mgx::set_error;

namespace {
inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
using mgx::validate;
}  // namespace

extern "C" {

// Host-only: test cases [pair_begin, pair_end) of `in` as a self-contained batch laid out in the caller's buffer.
int mgx_pairhmm_pack_batch(const mgx_pairhmm_input_t* in, uint64_t pair_begin, uint64_t pair_end, void* buf, size_t buf_bytes,
                           mgx_pairhmm_input_t* out, size_t* need) {
    int rc = validate(in);
    if (rc) return rc;
    if (!out || !need) { set_error("NULL argument"); return -EINVAL; }
    if (pair_begin > pair_end || pair_end > mgx::pack_n_pairs(in)) { set_error("test-case range outside the stream"); return -EINVAL; }
    mgx::PackPlan plan;
    const uint64_t bad = mgx::pack_plan(in, pair_begin, pair_end, &plan);
    if (bad) { set_error("test case %llu: index out of range", (unsigned long long)(pair_begin + bad - 1)); return -EINVAL; }
    const uint64_t n = pair_end - pair_begin, nr = plan.lread.size(), nh = plan.lhap.size(), rb = plan.roff.back(), hb = plan.hoff.back();
    size_t off = 0;
    const size_t o_roff = off; off = align_up(off + (nr + 1) * 8);
    const size_t o_hoff = off; off = align_up(off + (nh + 1) * 8);
    const size_t o_pr = off;   off = align_up(off + n * 4);
    const size_t o_ph = off;   off = align_up(off + n * 4);
    const size_t o_b = off;    off = align_up(off + rb);
    const size_t o_q = off;    off = align_up(off + rb);
    const size_t o_i = off;    off = align_up(off + rb);
    const size_t o_d = off;    off = align_up(off + rb);
    const size_t o_g = off;    off = align_up(off + rb);
    const size_t o_h = off;    off = align_up(off + hb);
    *need = off;
    if (!buf || buf_bytes < off) { set_error("buffer of %zu bytes needed", off); return -ENOSPC; }
    uint8_t* p = (uint8_t*)buf;
    memcpy(p + o_roff, plan.roff.data(), (nr + 1) * 8); memcpy(p + o_hoff, plan.hoff.data(), (nh + 1) * 8);
    memcpy(p + o_pr, plan.pair_read.data(), n * 4);     memcpy(p + o_ph, plan.pair_hap.data(), n * 4);
    mgx::pack_copy(in, plan, p + o_b, p + o_q, p + o_i, p + o_d, p + o_g, p + o_h);
    mgx_pairhmm_input_t o{};
    o.n_reads = nr; o.read_off = (const uint64_t*)(p + o_roff); o.bases = p + o_b; o.qual = p + o_q; o.ins = p + o_i; o.del = p + o_d; o.gcp = p + o_g;
    o.n_haps = nh; o.hap_off = (const uint64_t*)(p + o_hoff); o.hap_bases = p + o_h;
    o.n_pairs = n; o.pair_read = (const uint32_t*)(p + o_pr); o.pair_hap = (const uint32_t*)(p + o_ph);
    *out = o;
    return 0;
}


}  // extern "C"
