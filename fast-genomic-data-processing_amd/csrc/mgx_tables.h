// mgx_tables.h -- PairHMM probability tables (see mgx_tables.cpp).
#pragma once
#include <vector>

namespace mgx {

constexpr int kPh2prSize = 128;
constexpr int kMmSize = ((254 + 1) * (254 + 2)) >> 1;  // 32640, Context.h:50

template <typename T>
struct Tables {
    std::vector<T> ph2pr;  // [128]
    std::vector<T> mm;     // [32640] triangular matchToMatchProb
    std::vector<T> ph2pr_div3;  // [128] ph2pr / 3
    std::vector<T> gap_ratio;   // [128] ph2pr / (1 - ph2pr)
    T initial;             // 2^120 (float) / 2^1020 (double), Context.h:142,183
    T log10_initial;
};

template <typename T>
const Tables<T>& tables();
template <>
const Tables<float>& tables<float>();
template <>
const Tables<double>& tables<double>();

}  // namespace mgx
