// lbvh.hip.h -- the linear BVH builder of lbvh.h on the device: one kernel per step, rocPRIM's radix sort and prefix sum
// between them (sort.h).  Same element functions, same IEEE arithmetic -> the same keys, the same radix tree and the same
// output arrays as rayhip_lbvh::build_host (tests compare the two).  Included by rayhip.hip.
//
// Cost on MI355X, 3.0 M triangles in 486 k groups (leaf refinement of the Bistro-class scene): see DESIGN.md (N1).
#pragma once

#include <hip/hip_runtime.h>

#include "lbvh.h"
#include "sort.h"

namespace rayhip_lbvh {

constexpr int BUILD_BLOCK = 256;

__global__ void __launch_bounds__(BUILD_BLOCK) k_keys(const Box *__restrict__ prim_box, const uint32_t *__restrict__ prim_group,
                                                     const Box *__restrict__ group_centroids, const uint32_t n, uint64_t *__restrict__ keys,
                                                     uint32_t *__restrict__ order) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) {
        float c[3];
        centroid_of(prim_box[p], c);
        keys[p] = morton_key(prim_group[p], c, group_centroids[prim_group[p]]);
        order[p] = p;
    }
}

__global__ void __launch_bounds__(BUILD_BLOCK) k_hierarchy(const uint64_t *__restrict__ keys, const uint32_t n, RadixNode *__restrict__ rn) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 < n) {
        rn[i] = karras_node(keys, int(n), int(i));
    }
}
// (a second pass: a node's parent field is written by another thread than the node itself)
__global__ void __launch_bounds__(BUILD_BLOCK) k_parents(RadixNode *__restrict__ rn, const uint32_t n, uint32_t *__restrict__ node_parent,
                                                        uint32_t *__restrict__ leaf_parent) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 < n) {
        const uint32_t ch[2] = {rn[i].left, rn[i].right};
        for (int k = 0; k < 2; ++k) {
            if (ch[k] & LEAF_FLAG) {
                leaf_parent[ch[k] & ~LEAF_FLAG] = i;
            } else {
                node_parent[ch[k]] = i;
            }
        }
    }
}
// bottom-up boxes: one thread per primitive climbs; the second thread to arrive at a node finishes it (Karras 2012, sec. 4)
__global__ void __launch_bounds__(BUILD_BLOCK) k_fit(const Box *__restrict__ prim_box, const uint32_t *__restrict__ order,
                                                    const RadixNode *__restrict__ rn, const uint32_t *__restrict__ node_parent,
                                                    const uint32_t *__restrict__ leaf_parent, const uint32_t n, Box *__restrict__ nbox,
                                                    uint32_t *__restrict__ arrived) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n || n < 2) {
        return;
    }
    uint32_t cur = leaf_parent[p];
    while (cur != NONE) {
        __threadfence(); // what this thread wrote for the child below is visible before the arrival is counted
        if (atomicAdd(&arrived[cur], 1u) == 0u) {
            return; // first to arrive: the sibling's thread will finish this node
        }
        __threadfence();
        const uint32_t l = rn[cur].left, r = rn[cur].right;
        Box b = (l & LEAF_FLAG) ? prim_box[order[l & ~LEAF_FLAG]] : nbox[l];
        grow(b, (r & LEAF_FLAG) ? prim_box[order[r & ~LEAF_FLAG]] : nbox[r]);
        nbox[cur] = b;
        cur = node_parent[cur];
    }
}

// step 6: what every node / sorted position turns into; the three flag arrays are scanned into output indices
__global__ void __launch_bounds__(BUILD_BLOCK) k_kinds(const uint64_t *__restrict__ keys, const uint32_t n, const RadixNode *__restrict__ rn,
                                                      const uint32_t *__restrict__ node_parent, const uint32_t *__restrict__ leaf_parent,
                                                      const uint32_t leaf_max, const int roots_are_nodes, uint32_t *__restrict__ kind,
                                                      uint32_t *__restrict__ is_node, uint32_t *__restrict__ entry_count,
                                                      uint32_t *__restrict__ single_flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 < n) {
        uint32_t k;
        if (single_group_count(keys, rn[i].first, rn[i].last) == 0) {
            k = KIND_ABOVE;
        } else if (!range_is_leaf(keys, n, rn[i].first, rn[i].last, leaf_max, roots_are_nodes != 0)) {
            k = KIND_NODE;
        } else {
            const uint32_t par = node_parent[i];
            const bool parent_is_leaf = par != NONE && range_is_leaf(keys, n, rn[par].first, rn[par].last, leaf_max, roots_are_nodes != 0);
            k = parent_is_leaf ? KIND_INSIDE : KIND_LEAF;
        }
        kind[i] = k;
        is_node[i] = (k == KIND_NODE) ? 1u : 0u;
    }
    if (i < n) {
        const uint32_t par = n >= 2 ? leaf_parent[i] : NONE;
        const bool parent_is_leaf = par != NONE && range_is_leaf(keys, n, rn[par].first, rn[par].last, leaf_max, roots_are_nodes != 0);
        entry_count[i] = parent_is_leaf ? 1u : 2u; // a primitive that is a leaf of its own is written twice
        single_flag[i] = (roots_are_nodes && spans_a_group(keys, n, i, i)) ? 1u : 0u;
    }
}

struct EmitArgs {
    const uint64_t *keys;
    const uint32_t *order;
    const RadixNode *rn;
    const Box *prim_box, *nbox;
    const uint32_t *kind, *new_index, *entry_at, *single_index;
    uint32_t n, n_nodes_out;
    int leaf_is_primitive, roots_are_nodes;
    rayhip_bvh2_node *nodes;
    uint32_t *group_root, *entries;
};
__device__ inline void device_write_child(rayhip_bvh2_node &n, const int k, const Box &b, const uint32_t link) {
    if (k == 0) {
        n.ch_data0[0] = b.lo[0], n.ch_data0[1] = b.hi[0], n.ch_data0[2] = b.lo[1], n.ch_data0[3] = b.hi[1];
        n.ch_data2[0] = b.lo[2], n.ch_data2[1] = b.hi[2];
        n.left_child = link;
    } else {
        n.ch_data1[0] = b.lo[0], n.ch_data1[1] = b.hi[0], n.ch_data1[2] = b.lo[1], n.ch_data1[3] = b.hi[1];
        n.ch_data2[2] = b.lo[2], n.ch_data2[3] = b.hi[2];
        n.right_child = link;
    }
}
__device__ inline uint32_t device_link_of(const EmitArgs &a, const uint32_t c, Box &b) {
    uint32_t first, last;
    if (c & LEAF_FLAG) {
        first = last = (c & ~LEAF_FLAG);
        b = a.prim_box[a.order[first]];
    } else {
        first = a.rn[c].first, last = a.rn[c].last;
        b = a.nbox[c];
        if (a.kind[c] == KIND_NODE) {
            return a.new_index[c];
        }
    }
    if (a.leaf_is_primitive) {
        return (1u << 29) | a.order[first];
    }
    return leaf_word(a.entry_at[first], last - first + 1);
}
// step 7, one thread per radix-tree node and one per sorted position
__global__ void __launch_bounds__(BUILD_BLOCK) k_emit(const EmitArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i + 1 < a.n) {
        const uint32_t k = a.kind[i];
        const bool whole_group = spans_a_group(a.keys, a.n, a.rn[i].first, a.rn[i].last);
        if (k == KIND_NODE) {
            rayhip_bvh2_node o = {};
            Box b;
            const uint32_t l = device_link_of(a, a.rn[i].left, b);
            device_write_child(o, 0, b, l);
            const uint32_t r = device_link_of(a, a.rn[i].right, b);
            device_write_child(o, 1, b, r);
            a.nodes[a.new_index[i]] = o;
            if (whole_group) {
                a.group_root[group_of(a.keys[a.rn[i].first])] = a.new_index[i];
            }
        } else if (k == KIND_LEAF && whole_group) {
            a.group_root[group_of(a.keys[a.rn[i].first])] = leaf_word(a.entry_at[a.rn[i].first], a.rn[i].last - a.rn[i].first + 1);
        }
    }
    if (i < a.n) {
        const uint32_t e = a.entry_at[i], p = a.order[i];
        a.entries[e] = p;
        if (a.entry_at[i + 1] - e == 2) {
            a.entries[e + 1] = p;
        }
        if (spans_a_group(a.keys, a.n, i, i)) { // a group of one primitive
            const uint32_t w = a.leaf_is_primitive ? ((1u << 29) | p) : leaf_word(e, 1);
            if (a.roots_are_nodes) {
                rayhip_bvh2_node o = {};
                const Box b = a.prim_box[p];
                device_write_child(o, 0, b, w);
                if (a.leaf_is_primitive) {
                    const float far_ = 3.402823466e+38f;
                    device_write_child(o, 1, Box{{far_, far_, far_}, {far_, far_, far_}}, w);
                } else {
                    device_write_child(o, 1, b, w);
                }
                const uint32_t at = a.n_nodes_out + a.single_index[i];
                a.nodes[at] = o;
                a.group_root[group_of(a.keys[i])] = at;
            } else {
                a.group_root[group_of(a.keys[i])] = w;
            }
        }
    }
}

// device buffers of one build, sized once and reused (rayhip_ctx keeps one)
struct DeviceBuilder {
    hipStream_t stream = nullptr;
    std::vector<void *> owned;
    ~DeviceBuilder() {
        for (void *p : owned) {
            (void)hipFree(p);
        }
    }
    template <class T> T *alloc(const size_t count) {
        void *p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) {
            return nullptr;
        }
        owned.push_back(p);
        return static_cast<T *>(p);
    }
};

// Output identical to build_host(in).  Returns false with `why` set on a HIP / allocation failure.
inline bool build_device(hipStream_t stream, const Input &in, Output &out, std::string &why) {
    out = Output();
    out.group_root.assign(in.n_groups, NONE);
    out.bounds = empty_box();
    const uint32_t n = in.n_prims;
    if (n == 0) {
        return true;
    }
    {
        std::vector<Box> part(rayhip_host::host_threads() + 1, empty_box());
        const size_t per = (size_t(n) + part.size() - 1) / part.size();
        rayhip_host::parallel_blocks(part.size(), 1, [&](const size_t b, const size_t e) {
            for (size_t k = b; k < e; ++k) {
                for (size_t p = k * per; p < std::min<size_t>(n, (k + 1) * per); ++p) {
                    grow(part[k], in.prim_box[p]);
                }
            }
        });
        for (const Box &b : part) { // (min / max: the order of the parts does not matter)
            if (b.lo[0] <= b.hi[0]) {
                grow(out.bounds, b);
            }
        }
    }
    const std::vector<Box> cbox = centroid_boxes(in);
    DeviceBuilder B;
    B.stream = stream;
    const uint32_t n_int = n - 1;
#define LB_TRY(expr)                                                                                                    \
    do {                                                                                                                \
        const hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess) {                                                                                         \
            why = std::string(#expr) + ": " + hipGetErrorString(_e);                                                    \
            return false;                                                                                               \
        }                                                                                                               \
    } while (0)
    Box *d_box = B.alloc<Box>(n), *d_cbox = B.alloc<Box>(in.n_groups), *d_nbox = B.alloc<Box>(n_int);
    uint32_t *d_group = B.alloc<uint32_t>(n), *d_order0 = B.alloc<uint32_t>(n), *d_order = B.alloc<uint32_t>(n);
    uint64_t *d_keys0 = B.alloc<uint64_t>(n), *d_keys = B.alloc<uint64_t>(n);
    RadixNode *d_rn = B.alloc<RadixNode>(n_int);
    uint32_t *d_node_parent = B.alloc<uint32_t>(n_int), *d_leaf_parent = B.alloc<uint32_t>(n), *d_arrived = B.alloc<uint32_t>(n_int);
    uint32_t *d_kind = B.alloc<uint32_t>(n_int), *d_is_node = B.alloc<uint32_t>(n_int + 1), *d_new_index = B.alloc<uint32_t>(n_int + 1);
    uint32_t *d_entry_count = B.alloc<uint32_t>(n + 1), *d_entry_at = B.alloc<uint32_t>(n + 1);
    uint32_t *d_single = B.alloc<uint32_t>(n + 1), *d_single_index = B.alloc<uint32_t>(n + 1);
    uint32_t *d_group_root = B.alloc<uint32_t>(in.n_groups);
    if (!d_box || !d_cbox || !d_nbox || !d_group || !d_order0 || !d_order || !d_keys0 || !d_keys || !d_rn || !d_node_parent || !d_leaf_parent ||
        !d_arrived || !d_kind || !d_is_node || !d_new_index || !d_entry_count || !d_entry_at || !d_single || !d_single_index || !d_group_root) {
        why = "out of device memory in the BVH builder";
        return false;
    }
    size_t sort_bytes = 0, scan_bytes = 0;
    LB_TRY(rt::sort_pairs_u64(nullptr, &sort_bytes, d_keys0, d_keys, d_order0, d_order, n, stream));
    LB_TRY(rt::exclusive_scan_u32(nullptr, &scan_bytes, d_entry_count, d_entry_at, size_t(n) + 1, stream));
    void *d_temp = B.alloc<uint8_t>(std::max(sort_bytes, scan_bytes));
    if (!d_temp) {
        why = "out of device memory in the BVH builder";
        return false;
    }
    const int g_n = int((n + BUILD_BLOCK - 1) / BUILD_BLOCK), g_int = std::max(1, int((n_int + BUILD_BLOCK - 1) / BUILD_BLOCK));
    LB_TRY(hipMemcpyAsync(d_box, in.prim_box, size_t(n) * sizeof(Box), hipMemcpyHostToDevice, stream));
    LB_TRY(hipMemcpyAsync(d_group, in.prim_group, size_t(n) * 4, hipMemcpyHostToDevice, stream));
    LB_TRY(hipMemcpyAsync(d_cbox, cbox.data(), size_t(in.n_groups) * sizeof(Box), hipMemcpyHostToDevice, stream));
    LB_TRY(hipMemsetAsync(d_node_parent, 0xff, size_t(std::max(n_int, 1u)) * 4, stream));
    LB_TRY(hipMemsetAsync(d_leaf_parent, 0xff, size_t(n) * 4, stream));
    LB_TRY(hipMemsetAsync(d_arrived, 0, size_t(std::max(n_int, 1u)) * 4, stream));
    LB_TRY(hipMemsetAsync(d_group_root, 0xff, size_t(in.n_groups) * 4, stream));
    LB_TRY(hipMemsetAsync(d_is_node, 0, size_t(n_int + 1) * 4, stream));
    LB_TRY(hipMemsetAsync(d_entry_count, 0, size_t(n + 1) * 4, stream));
    LB_TRY(hipMemsetAsync(d_single, 0, size_t(n + 1) * 4, stream));
    // 2-3: keys, sort
    k_keys<<<g_n, BUILD_BLOCK, 0, stream>>>(d_box, d_group, d_cbox, n, d_keys0, d_order0);
    LB_TRY(rt::sort_pairs_u64(d_temp, &sort_bytes, d_keys0, d_keys, d_order0, d_order, n, stream));
    // 4-5: hierarchy, fit
    if (n_int > 0) {
        k_hierarchy<<<g_int, BUILD_BLOCK, 0, stream>>>(d_keys, n, d_rn);
        k_parents<<<g_int, BUILD_BLOCK, 0, stream>>>(d_rn, n, d_node_parent, d_leaf_parent);
        k_fit<<<g_n, BUILD_BLOCK, 0, stream>>>(d_box, d_order, d_rn, d_node_parent, d_leaf_parent, n, d_nbox, d_arrived);
    }
    // 6: kinds + the three scans
    k_kinds<<<g_n, BUILD_BLOCK, 0, stream>>>(d_keys, n, d_rn, d_node_parent, d_leaf_parent, in.leaf_max, in.roots_are_nodes ? 1 : 0, d_kind,
                                             d_is_node, d_entry_count, d_single);
    LB_TRY(rt::exclusive_scan_u32(d_temp, &scan_bytes, d_is_node, d_new_index, size_t(n_int) + 1, stream));
    LB_TRY(rt::exclusive_scan_u32(d_temp, &scan_bytes, d_entry_count, d_entry_at, size_t(n) + 1, stream));
    LB_TRY(rt::exclusive_scan_u32(d_temp, &scan_bytes, d_single, d_single_index, size_t(n) + 1, stream));
    uint32_t n_nodes = 0, n_entries = 0, n_single = 0;
    LB_TRY(hipMemcpyAsync(&n_nodes, d_new_index + n_int, 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(&n_entries, d_entry_at + n, 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(&n_single, d_single_index + n, 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipStreamSynchronize(stream));
    // 7: emit
    rayhip_bvh2_node *d_nodes = B.alloc<rayhip_bvh2_node>(size_t(n_nodes) + n_single);
    uint32_t *d_entries = B.alloc<uint32_t>(n_entries);
    if (!d_nodes || !d_entries) {
        why = "out of device memory in the BVH builder";
        return false;
    }
    EmitArgs a;
    a.keys = d_keys, a.order = d_order, a.rn = d_rn, a.prim_box = d_box, a.nbox = d_nbox;
    a.kind = d_kind, a.new_index = d_new_index, a.entry_at = d_entry_at, a.single_index = d_single_index;
    a.n = n, a.n_nodes_out = n_nodes, a.leaf_is_primitive = in.leaf_is_primitive ? 1 : 0, a.roots_are_nodes = in.roots_are_nodes ? 1 : 0;
    a.nodes = d_nodes, a.group_root = d_group_root, a.entries = d_entries;
    k_emit<<<g_n, BUILD_BLOCK, 0, stream>>>(a);
    LB_TRY(hipGetLastError());
    out.nodes.resize(size_t(n_nodes) + n_single);
    out.entries.resize(n_entries);
    LB_TRY(hipMemcpyAsync(out.nodes.data(), d_nodes, out.nodes.size() * sizeof(rayhip_bvh2_node), hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(out.entries.data(), d_entries, size_t(n_entries) * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipMemcpyAsync(out.group_root.data(), d_group_root, size_t(in.n_groups) * 4, hipMemcpyDeviceToHost, stream));
    LB_TRY(hipStreamSynchronize(stream));
#undef LB_TRY
    return true;
}

} // namespace rayhip_lbvh
