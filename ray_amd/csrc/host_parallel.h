// host_parallel.h -- a blocked parallel-for over the host cores, for the once-per-upload host passes (leaf gathering of the
// refinement, splice, centroid boxes).  Threads are created per call (tens of microseconds each; the loops they share out run
// for tens of milliseconds); the number respects the process's affinity mask and RAYHIP_HOST_THREADS.
#pragma once

#include <algorithm>
#include <cstdlib>
#include <thread>
#include <vector>

#if defined(__linux__)
#include <sched.h>
#endif

namespace rayhip_host {

inline unsigned host_threads() {
    static const unsigned n = [] {
        if (const char *e = getenv("RAYHIP_HOST_THREADS")) {
            return unsigned(std::max(1, atoi(e)));
        }
        unsigned hw = std::thread::hardware_concurrency();
#if defined(__linux__)
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) {
            hw = std::min<unsigned>(hw ? hw : 1u, unsigned(CPU_COUNT(&set)));
        }
#endif
        return std::max(1u, std::min(hw, 16u)); // (containers often grant far fewer cores than the machine shows)
    }();
    return n;
}

// fn(begin, end) over [0, n) in contiguous blocks, one per thread; serial below `grain` items per thread
template <class F> inline void parallel_blocks(const size_t n, const size_t grain, F &&fn) {
    const size_t t = std::min<size_t>(host_threads(), grain ? std::max<size_t>(1, n / grain) : 1);
    if (t <= 1 || n == 0) {
        fn(size_t(0), n);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve(t - 1);
    const size_t per = (n + t - 1) / t;
    for (size_t k = 1; k < t; ++k) {
        const size_t b = std::min(n, k * per), e = std::min(n, (k + 1) * per);
        if (b < e) {
            pool.emplace_back([&fn, b, e] { fn(b, e); });
        }
    }
    fn(size_t(0), std::min(n, per));
    for (std::thread &th : pool) {
        th.join();
    }
}

} // namespace rayhip_host
