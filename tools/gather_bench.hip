// gather_bench.hip -- what does a divergent 64-byte node fetch cost on MI355X?  (tuning tool, not product code)
//
// BVH traversal is a per-lane dependent chain of 64-byte node reads at unrelated addresses.  This microbenchmark
// measures the node fetch rate of that access pattern in isolation, for the two ways a wavefront can issue it:
//
//   lane  : every lane reads its own node with four 16-byte loads          (4 L1 accesses per node)
//   quad  : the four lanes of a quad read one node together, 16 bytes each (1 L1 access per node if the texture
//           addresser coalesces a quad), four rounds + a 4x4 transpose through DPP to hand every lane its node
//
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o /tmp/gather_bench && /tmp/gather_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                                                        \
    do {                                                                                                                \
        hipError_t e = (x);                                                                                             \
        if (e != hipSuccess) {                                                                                          \
            printf("%s failed: %s\n", #x, hipGetErrorString(e));                                                        \
            return 1;                                                                                                   \
        }                                                                                                               \
    } while (0)

__device__ inline uint32_t mix(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// every lane: 4 x dwordx4 of its own node
template <int WAVES>
__global__ __launch_bounds__(64, WAVES) void k_lane(const float4 *nodes, uint32_t mask, int iters, uint32_t *out, float frac_active) {
    const uint32_t tid = blockIdx.x * 64 + threadIdx.x;
    uint32_t cur = mix(tid) & mask;
    float acc = 0.0f;
    const bool active = (mix(tid * 7u + 1u) & 0xffffu) < uint32_t(frac_active * 65536.0f);
    if (active) {
        for (int i = 0; i < iters; ++i) {
            const float4 *p = nodes + size_t(cur) * 4;
            const float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x + b.y + c.z;
            cur = (__float_as_uint(d.x) + mix(cur + i)) & mask; // dependent on the loaded data
        }
    }
    out[tid] = cur + uint32_t(acc);
}

__device__ inline uint32_t quad_bcast(uint32_t v, int k) {
    switch (k) {
    case 0: return __builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true);
    case 1: return __builtin_amdgcn_mov_dpp(v, 0x55, 0xf, 0xf, true);
    case 2: return __builtin_amdgcn_mov_dpp(v, 0xaa, 0xf, 0xf, true);
    default: return __builtin_amdgcn_mov_dpp(v, 0xff, 0xf, 0xf, true);
    }
}
__device__ inline float xor1(float v) { return __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0xb1, 0xf, 0xf, true)); } // quad_perm [1,0,3,2]
__device__ inline float xor2(float v) { return __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), 0x4e, 0xf, 0xf, true)); } // quad_perm [2,3,0,1]

typedef float v4f __attribute__((ext_vector_type(4)));

// 4x4 transpose of 16-byte "registers x lanes-in-quad": on entry r_i of lane k = chunk k of node i; on exit
// r_i of lane k = chunk i of node k.  Two butterfly stages; each moves half of the data through DPP.
#define XCH(A, B, XF, BIT)                                                                                               \
    {                                                                                                                   \
        const v4f send = BIT ? A : B; /* what the partner needs from me */                                              \
        v4f got;                                                                                                        \
        got.x = XF(send.x), got.y = XF(send.y), got.z = XF(send.z), got.w = XF(send.w);                                 \
        A = BIT ? got : A;                                                                                              \
        B = BIT ? B : got;                                                                                              \
    }

// predicated 16-byte load the compiler cannot serialise: issued inside the branch, waited for once, later
__device__ inline void load16_if(v4f &dst, const float4 *p, bool pred) {
    if (pred) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
    }
}

// quad-cooperative: 4 rounds, each lane loads 16 bytes, then transpose
template <int WAVES>
__global__ __launch_bounds__(64, WAVES) void k_quad(const float4 *nodes, uint32_t mask, int iters, uint32_t *out, float frac_active) {
    const uint32_t tid = blockIdx.x * 64 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 3u;
    const bool b0 = lane & 1u, b1 = lane & 2u;
    uint32_t cur = mix(tid) & mask;
    float acc = 0.0f;
    const bool active = (mix(tid * 7u + 1u) & 0xffffu) < uint32_t(frac_active * 65536.0f);
    for (int i = 0; i < iters; ++i) {
        const uint32_t want = active ? cur : 0xffffffffu;
        const uint32_t n0 = quad_bcast(want, 0), n1 = quad_bcast(want, 1), n2 = quad_bcast(want, 2), n3 = quad_bcast(want, 3);
        v4f r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0;
        load16_if(r0, nodes + size_t(n0) * 4 + lane, n0 != 0xffffffffu);
        load16_if(r1, nodes + size_t(n1) * 4 + lane, n1 != 0xffffffffu);
        load16_if(r2, nodes + size_t(n2) * 4 + lane, n2 != 0xffffffffu);
        load16_if(r3, nodes + size_t(n3) * 4 + lane, n3 != 0xffffffffu);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
        XCH(r0, r1, xor1, b0)
        XCH(r2, r3, xor1, b0)
        XCH(r0, r2, xor2, b1)
        XCH(r1, r3, xor2, b1)
        if (active) {
            acc += r0.x + r1.y + r2.z;
            cur = (__float_as_uint(r3.x) + mix(cur + i)) & mask;
        }
    }
    out[tid] = cur + uint32_t(acc);
}

// one 16-byte load per lane per step (lower bound on "accesses" cost)
template <int WAVES>
__global__ __launch_bounds__(64, WAVES) void k_lane16(const float4 *nodes, uint32_t mask, int iters, uint32_t *out, float frac_active) {
    const uint32_t tid = blockIdx.x * 64 + threadIdx.x;
    uint32_t cur = mix(tid) & mask;
    const bool active = (mix(tid * 7u + 1u) & 0xffffu) < uint32_t(frac_active * 65536.0f);
    if (active) {
        for (int i = 0; i < iters; ++i) {
            const float4 d = nodes[size_t(cur) * 4 + 3];
            cur = (__float_as_uint(d.x) + mix(cur + i)) & mask;
        }
    }
    out[tid] = cur;
}

template <class F> static float time_ms(F &&launch, int reps = 3) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    const int iters = 256;
    const int waves_total = 256 * 4 * 8; // one full residency at 8 waves/SIMD
    uint32_t *out;
    CHECK(hipMalloc(&out, size_t(waves_total) * 64 * 4 * 4));
    printf("%-10s %8s %6s %6s | %10s %10s\n", "kernel", "table", "w/SIMD", "active", "Gnodes/s", "GB/s(64B)");
    for (int log_nodes : {14, 18, 21, 23}) { // 1 MiB (L2 resident), 16 MiB, 128 MiB (MALL), 512 MiB (HBM)
        const size_t n_nodes = size_t(1) << log_nodes;
        float4 *nodes;
        CHECK(hipMalloc(&nodes, n_nodes * 64));
        std::vector<uint32_t> h(n_nodes * 16);
        uint32_t s = 12345u;
        for (auto &v : h) {
            s = s * 1664525u + 1013904223u;
            v = s >> 4;
        }
        CHECK(hipMemcpy(nodes, h.data(), n_nodes * 64, hipMemcpyHostToDevice));
        const uint32_t mask = uint32_t(n_nodes - 1);
        for (float frac : {1.0f, 0.3f}) {
            for (int w : {4, 8}) {
                const int blocks = 256 * 4 * w * 2;
                auto report = [&](const char *name, float ms) {
                    const double nodes_done = double(blocks) * 64 * frac * iters;
                    printf("%-10s %6zuMB %6d %6.1f | %10.2f %10.1f\n", name, n_nodes * 64 >> 20, w, frac, nodes_done / ms * 1e-6,
                           nodes_done * 64 / ms * 1e-6);
                };
                if (w == 4) {
                    report("lane", time_ms([&] { k_lane<4><<<blocks, 64>>>(nodes, mask, iters, out, frac); }));
                    report("quad", time_ms([&] { k_quad<4><<<blocks, 64>>>(nodes, mask, iters, out, frac); }));
                    report("lane16", time_ms([&] { k_lane16<4><<<blocks, 64>>>(nodes, mask, iters, out, frac); }));
                } else {
                    report("lane", time_ms([&] { k_lane<8><<<blocks, 64>>>(nodes, mask, iters, out, frac); }));
                    report("quad", time_ms([&] { k_quad<8><<<blocks, 64>>>(nodes, mask, iters, out, frac); }));
                    report("lane16", time_ms([&] { k_lane16<8><<<blocks, 64>>>(nodes, mask, iters, out, frac); }));
                }
            }
        }
        CHECK(hipFree(nodes));
    }
    return 0;
}
