// bvh4_build.hip.h -- the 4-wide collapse of bvh4_build.h on the device: what rayhip_scene_upload runs (the host loop over
// 650 k wide nodes of the Bistro-class scene took 230 ms of a 0.70 s upload; here: level by level, one thread per wide node).
//
// Level-synchronous: the frontier of a level holds (BVH2 node, index of the wide node that stands for it); a thread decides
// its node's children with the element functions of bvh4_build.h (the same IEEE operations as the host driver, so the same
// tree), takes consecutive indices for its inner children from one counter -- the children of a node stay next to each other --
// writes its node and appends the children to the next frontier.  The node ORDER is breadth-first instead of the host driver's
// depth-first one (measured irrelevant: profiles/r01/K2_findings_round1.md); topology, quantised boxes and therefore every
// visit counter are identical (tests/test_gpu_bvh_build.py).
#pragma once

#include <hip/hip_runtime.h>

#include "bvh4_build.h"

namespace rayhip_bvh4 {

struct Frontier {
    uint32_t bvh2_node, out_index;
};

__global__ void __launch_bounds__(256) k_collapse_level(const rayhip_bvh2_node *__restrict__ nodes, const Frontier *__restrict__ in, const uint32_t n_in,
                                                       rt::Bvh4Node *__restrict__ out_nodes, Frontier *__restrict__ next, const uint32_t next_base,
                                                       uint32_t *__restrict__ counter /* nodes allocated so far */, uint32_t *__restrict__ failed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_in) {
        return;
    }
    const Frontier f = in[i];
    Slot slots[4];
    const int n_slots = wide_children(nodes, f.bvh2_node, slots);
    rt::Bvh4Node node;
    if (!quantise(slots, n_slots, node)) {
        atomicExch(failed, 1u);
        return;
    }
    uint32_t n_inner = 0;
    for (int c = 0; c < n_slots; ++c) {
        n_inner += is_leaf(slots[c].ref) ? 0u : 1u;
    }
    const uint32_t first_child = n_inner ? atomicAdd(counter, n_inner) : 0u;
    uint32_t k = 0;
    for (int c = 0; c < n_slots; ++c) {
        if (is_leaf(slots[c].ref)) {
            node.child[c] = slots[c].ref;
        } else {
            node.child[c] = first_child + k;
            next[first_child + k - next_base] = Frontier{slots[c].ref, first_child + k};
            ++k;
        }
    }
    out_nodes[f.out_index] = node;
}

// `d_nodes`: the BVH2 as uploaded (device); `roots`: the distinct BLAS roots (BVH2 node indices).  Fills `d_out` (capacity
// n_nodes2 wide nodes: a wide node stands for at least one BVH2 node) and returns the number of wide nodes; the wide node of
// roots[r] is r.  Returns false on a device error or if a box could not be quantised.
// Returns false on failure with the reason in `why`; `*unquantisable` (optional) tells the one failure that is a property of the scene,
// not an error: a child box that the 8-bit grid cannot hold (non-finite or astronomically large extent) -- the caller then walks the BVH2.
inline bool build_device(hipStream_t stream, const rayhip_bvh2_node *d_nodes, const uint32_t n_nodes2, const std::vector<uint32_t> &roots,
                         rt::Bvh4Node *d_out, uint32_t &out_count, std::string &why, bool *unquantisable = nullptr) {
    out_count = 0;
    if (unquantisable) {
        *unquantisable = false;
    }
    if (roots.empty()) {
        return true;
    }
    Frontier *d_front[2] = {nullptr, nullptr};
    uint32_t *d_ctrl = nullptr; // [0] node counter, [1] failure flag
#define B4_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        const hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) {                                                                                        \
            why = std::string(#expr) + ": " + hipGetErrorString(e_);                                                   \
            (void)hipFree(d_front[0]), (void)hipFree(d_front[1]), (void)hipFree(d_ctrl);                               \
            return false;                                                                                              \
        }                                                                                                              \
    } while (0)
    B4_TRY(hipMalloc(&d_front[0], size_t(n_nodes2) * sizeof(Frontier)));
    B4_TRY(hipMalloc(&d_front[1], size_t(n_nodes2) * sizeof(Frontier)));
    B4_TRY(hipMalloc(&d_ctrl, 2 * sizeof(uint32_t)));
    std::vector<Frontier> first(roots.size());
    for (size_t r = 0; r < roots.size(); ++r) {
        first[r] = Frontier{roots[r], uint32_t(r)};
    }
    uint32_t ctrl[2] = {uint32_t(roots.size()), 0u};
    B4_TRY(hipMemcpyAsync(d_front[0], first.data(), first.size() * sizeof(Frontier), hipMemcpyHostToDevice, stream));
    B4_TRY(hipMemcpyAsync(d_ctrl, ctrl, sizeof(ctrl), hipMemcpyHostToDevice, stream));
    uint32_t level_base = 0, level_count = uint32_t(roots.size());
    for (int level = 0; level_count != 0; ++level) {
        if (level > 4096) {
            why = "the BVH is deeper than any tree";
            (void)hipFree(d_front[0]), (void)hipFree(d_front[1]), (void)hipFree(d_ctrl);
            return false;
        }
        const uint32_t next_base = level_base + level_count;
        k_collapse_level<<<(level_count + 255) / 256, 256, 0, stream>>>(d_nodes, d_front[level & 1], level_count, d_out, d_front[(level + 1) & 1],
                                                                        next_base, d_ctrl, d_ctrl + 1);
        B4_TRY(hipGetLastError());
        B4_TRY(hipMemcpyAsync(ctrl, d_ctrl, sizeof(ctrl), hipMemcpyDeviceToHost, stream));
        B4_TRY(hipStreamSynchronize(stream));
        if (ctrl[1]) {
            why = "a child box cannot be quantised (non-finite coordinates)";
            if (unquantisable) {
                *unquantisable = true;
            }
            (void)hipFree(d_front[0]), (void)hipFree(d_front[1]), (void)hipFree(d_ctrl);
            return false;
        }
        if (ctrl[0] > n_nodes2) {
            why = "more wide nodes than BVH2 nodes: not a tree";
            (void)hipFree(d_front[0]), (void)hipFree(d_front[1]), (void)hipFree(d_ctrl);
            return false;
        }
        level_base = next_base, level_count = ctrl[0] - next_base;
    }
#undef B4_TRY
    out_count = ctrl[0];
    (void)hipFree(d_front[0]), (void)hipFree(d_front[1]), (void)hipFree(d_ctrl);
    return true;
}

} // namespace rayhip_bvh4
