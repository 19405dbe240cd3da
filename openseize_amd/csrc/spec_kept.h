// The tables of the last few (taps, cascade, tolerance) per table kind are kept for the process:
// the same filters run over recording after recording, and 3-5 ms of long-double arithmetic per
// table are 3-4 % each of a 1e8-sample stream of 256 channels (spec_tables.h builds them).
#pragma once
#include <mutex>
#include <utility>
#include <vector>

#include "spec_tables.h"

namespace osz {
namespace spec {

enum { kKeptZpn = 0, kKeptSpecn = 1, kKeptKinds = 2 };

template <class Build>
inline TablesZp kept_tables(int kind, const std::vector<double> &taps, const double *coef, int nsec, double tol,
                            bool forgets, Build build) {
    std::vector<double> key(taps);
    key.insert(key.end(), coef, coef + 6 * (size_t)nsec);
    key.push_back(tol);
    key.push_back(forgets ? 1.0 : 0.0);
    static std::mutex mu;
    static std::vector<std::pair<std::vector<double>, TablesZp>> kept[kKeptKinds];
    {
        std::lock_guard<std::mutex> lock(mu);
        for (const auto &e : kept[kind])
            if (e.first == key) return e.second;
    }
    TablesZp T = build();
    std::lock_guard<std::mutex> lock(mu);
    if (kept[kind].size() >= 8) kept[kind].erase(kept[kind].begin());
    kept[kind].emplace_back(std::move(key), T);
    return T;
}

}  // namespace spec
}  // namespace osz
