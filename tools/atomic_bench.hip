// atomic_bench.hip -- cost of the wave-aggregated slot allocation (one atomicAdd per wavefront) as a function of how
// many distinct counters the wavefronts are spread over.  (tuning tool, not product code)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__global__ __launch_bounds__(64) void k_alloc(uint32_t *counters, int k, int stride_words, int per_wave, uint32_t *out) {
    const uint32_t wave = blockIdx.x;
    uint32_t acc = 0;
    for (int i = 0; i < per_wave; ++i) {
        uint32_t base = 0;
        if (threadIdx.x == 0) {
            base = atomicAdd(&counters[size_t((wave + i) % k) * stride_words], 64u);
        }
        base = __shfl(base, 0);
        acc += base;
    }
    out[wave * 64 + threadIdx.x] = acc;
}

int main() {
    uint32_t *counters, *out;
    const int waves = 32768;
    if (hipMalloc(&counters, 1 << 22) != hipSuccess || hipMalloc(&out, size_t(waves) * 64 * 4) != hipSuccess) {
        return 1;
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    printf("%8s %8s %8s | %10s %12s\n", "counters", "stride", "per_wave", "ms", "ns/atomic");
    for (int per_wave : {1, 4}) {
        for (int stride : {16, 64, 1024}) { // words: 64 B, 256 B, 4 KiB
            for (int k : {1, 2, 4, 8, 16, 32, 64, 256}) {
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) {
                    (void)hipMemsetAsync(counters, 0, 1 << 22);
                    (void)hipEventRecord(e0);
                    k_alloc<<<waves, 64>>>(counters, k, stride, per_wave, out);
                    (void)hipEventRecord(e1);
                    (void)hipEventSynchronize(e1);
                    float ms;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
                printf("%8d %7dB %8d | %10.4f %12.2f\n", k, stride * 4, per_wave, best, best * 1e6 / (double(waves) * per_wave));
            }
        }
    }
    return 0;
}
