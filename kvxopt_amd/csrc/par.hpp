// Host threads for the symbolic analysis: a plain fork-join over index ranges (no OpenMP runtime in the library).
// Exceptions of a worker are carried to the caller (abi_guard.hpp maps them to status codes at the C ABI).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <exception>
#include <thread>
#include <vector>

namespace kvx {

inline int analyze_threads()
{
    int t = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("KVX_ANALYZE_THREADS")) t = std::max(1, std::min(64, atoi(e)));
    return t;
}

// fn(lo, hi) on contiguous chunks of [0, n); chunk boundaries depend on n and the thread count only.  The result of every use in
// this library is independent of the chunking (disjoint outputs, or outputs made canonical afterwards).
template <class Fn>
void parallel_for(int64_t n, int nthreads, int64_t min_chunk, Fn fn)
{
    if (n <= 0) return;
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(nthreads, n / std::max<int64_t>(min_chunk, 1)));
    if (T <= 1) { fn((int64_t)0, n); return; }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err((size_t)T);
    auto run = [&](int t) {
        try { fn(n * t / T, n * (t + 1) / T); } catch (...) { err[(size_t)t] = std::current_exception(); }
    };
    try {
        for (int t = 1; t < T; t++) th.emplace_back(run, t);
    } catch (...) {                                  // could not start a thread: do its share here
        for (int t = (int)th.size() + 1; t < T; t++) run(t);
    }
    run(0);
    for (auto &x : th) x.join();
    for (auto &e : err)
        if (e) std::rethrow_exception(e);
}

}  // namespace kvx
