// mgx_tables.cpp -- host-side construction of the PairHMM probability tables (product code).
//
// The device kernels look transition/emission probabilities up in two tables per precision.
// To return the reference's numbers the tables have to be the reference's tables, including
// its deliberate quirks, so they are built here the way Context<NUMBER> builds them
// (deepmutect/Mutect2Cpp-master/src/intel/pairhmm/Context.h):
//   * ph2pr[q] = 10^(-q/10), q in [0,127]; fp32 uses powf on float operands   (:139, :180)
//   * matchToMatchProb[(max*(max+1)/2) + min] through the Jacobian-logarithm approximation
//     with a 1e-4 step table of 80 001 entries, hard rounding, and the TRUNCATED constant
//     INV_LN10 = 0.434294                                                     (:65-122)
// Quality bytes are masked to 7 bits before lookup (ReadForPairHMM.cpp:34-36), so only the
// first 128*129/2 = 8256 triangular entries are reachable; the whole 255-quality triangle is
// still built so that index arithmetic is identical.
#include "mgx_tables.h"

#include <algorithm>
#include <cmath>
#include <mutex>
#include <vector>

namespace mgx {
namespace {

constexpr int kMaxQual = 254;
constexpr double kJacTol = 8.0;
constexpr double kJacStep = 0.0001;
constexpr double kJacInvStep = 1.0 / kJacStep;
constexpr int kJacSize = 80001;

template <typename T>
struct Builder {
    std::vector<T> jac;

    static int fast_round(T d) { return (d > T(0)) ? int(d + T(0.5)) : int(d - T(0.5)); }

    T approx_log10_sum(T small, T big) const {
        if (small > big) std::swap(small, big);
        if (std::isinf(small) || std::isinf(big)) return big;
        T diff = big - small;
        if (diff >= T(kJacTol)) return big;
        int ind = fast_round(T(diff * T(kJacInvStep)));
        return big + jac[ind];
    }

    void build(Tables<T>& t) {
        jac.resize(kJacSize);
        for (int k = 0; k < kJacSize; ++k)
            jac[k] = T(std::log10(1.0 + std::pow(10.0, -double(k) * kJacStep)));
        const double inv_ln10 = 0.434294;  // sic: the reference truncates 1/ln(10)
        t.mm.resize(kMmSize);
        for (int i = 0, offset = 0; i <= kMaxQual; offset += ++i)
            for (int j = 0; j <= i; ++j) {
                double log10_sum = approx_log10_sum(T(-0.1) * T(i), T(-0.1) * T(j));
                double m2m_log10 = std::log1p(-std::min(1.0, std::pow(10, log10_sum))) * inv_ln10;
                t.mm[offset + j] = T(std::pow(10, m2m_log10));
            }
    }
};

}  // namespace

template <>
const Tables<float>& tables<float>() {
    static Tables<float> t;
    static std::once_flag once;
    std::call_once(once, [] {
        Builder<float>().build(t);
        t.ph2pr.resize(kPh2prSize);
        for (int x = 0; x < kPh2prSize; ++x) t.ph2pr[x] = powf(10.f, -float(x) / 10.f);
        t.ph2pr_div3.resize(kPh2prSize); t.gap_ratio.resize(kPh2prSize);
        for (int x = 0; x < kPh2prSize; ++x) {
            t.ph2pr_div3[x] = t.ph2pr[x] / 3.0f;                                   // VEC_DIV(distm, 3), avx-pairhmm-template.h:150
            const float g = 1.0f - t.ph2pr[x];
            t.gap_ratio[x] = g != 0.0f ? t.ph2pr[x] / g : 0.0f;
        }
        t.initial = ldexpf(1.f, 120);
        t.log10_initial = log10f(t.initial);
    });
    return t;
}

template <>
const Tables<double>& tables<double>() {
    static Tables<double> t;
    static std::once_flag once;
    std::call_once(once, [] {
        Builder<double>().build(t);
        t.ph2pr.resize(kPh2prSize);
        for (int x = 0; x < kPh2prSize; ++x) t.ph2pr[x] = std::pow(10.0, -double(x) / 10.0);
        t.ph2pr_div3.resize(kPh2prSize); t.gap_ratio.resize(kPh2prSize);
        for (int x = 0; x < kPh2prSize; ++x) {
            t.ph2pr_div3[x] = t.ph2pr[x] / 3.0;
            const double g = 1.0 - t.ph2pr[x];
            t.gap_ratio[x] = g != 0.0 ? t.ph2pr[x] / g : 0.0;
        }
        t.initial = std::ldexp(1.0, 1020);
        t.log10_initial = std::log10(t.initial);
    });
    return t;
}

}  // namespace mgx
