// bvh4_build.h -- collapse of the reference's BVH2 BLAS trees into the 4-wide quantised form of rt_bvh4.h: the element functions
// (which children a wide node gets, how their boxes are quantised -- host and device, the same IEEE operations) and the host
// driver over them (tests/hostsim, RAYHIP_BVH_BUILD_ON_HOST=1).  The device driver is bvh4_build.hip.h: what a scene upload runs.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "rt_bvh4.h"

namespace rayhip_bvh4 {

struct Box {
    float lo[3], hi[3];
};
struct Slot {
    Box box;
    uint32_t ref; // BVH2 child word: inner node index or leaf word
};

RT_HD bool is_leaf(uint32_t w) { return (w & rt::BVH2_PRIM_COUNT_BITS) != 0; }

// the two child boxes stored in a bvh2 node (reference layout, Core.h:72-80: ch_data0 = child 0 {xmin,xmax,ymin,ymax},
// ch_data1 = child 1, ch_data2 = {z0min,z0max,z1min,z1max})
RT_HD void children_of(const rayhip_bvh2_node &n, Slot out[2]) {
    out[0].box = Box{{n.ch_data0[0], n.ch_data0[2], n.ch_data2[0]}, {n.ch_data0[1], n.ch_data0[3], n.ch_data2[1]}};
    out[1].box = Box{{n.ch_data1[0], n.ch_data1[2], n.ch_data2[2]}, {n.ch_data1[1], n.ch_data1[3], n.ch_data2[3]}};
    out[0].ref = n.left_child, out[1].ref = n.right_child;
}

RT_HD float half_area(const Box &b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// plane q of the node grid as a real number: org and q * step (a power of two times an 8-bit integer) are fp32 values, so
// their sum is exact in double
RT_HD double real_plane(const uint32_t q, const float step, const float org) { return double(org) + double(q) * double(step); }

// Quantise `n_slots` child boxes onto the node grid.  Returns false if a box cannot be represented conservatively
// (non-finite coordinates): the caller then keeps the BVH2 for the whole scene.
RT_HD bool quantise(const Slot *slots, const int n_slots, rt::Bvh4Node &out) {
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = slots[0].box.lo[a], hi[a] = slots[0].box.hi[a];
        for (int c = 1; c < n_slots; ++c) {
            lo[a] = fminf(lo[a], slots[c].box.lo[a]);
            hi[a] = fmaxf(hi[a], slots[c].box.hi[a]);
        }
        if (!(fabsf(lo[a]) <= 3.402823466e+38f) || !(fabsf(hi[a]) <= 3.402823466e+38f) || hi[a] < lo[a]) { // (NaN or infinite)
            return false;
        }
    }
    out = rt::Bvh4Node{};
    float scale[3];
    for (int a = 0; a < 3; ++a) {
        out.org[a] = lo[a];
        // smallest power of two s with org + 255 * s >= hi, evaluated with the device's de-quantisation
        int e = 1;
        const float need = (hi[a] - lo[a]) / 255.0f;
        if (need > 0.0f) {
            int ex;
            (void)frexpf(need, &ex); // need = m * 2^ex, m in [0.5, 1)
            e = ex + 127;            // 2^ex >= need
            if (e < 1) {
                e = 1;
            }
        }
        while (e < 254 && real_plane(255u, rt::uint_as_float(uint32_t(e) << 23), lo[a]) < double(hi[a])) {
            ++e;
        }
        if (e >= 254) {
            return false;
        }
        scale[a] = rt::uint_as_float(uint32_t(e) << 23);
    }
    out.step_x = scale[0], out.step_y = scale[1], out.step_z = scale[2];
    for (int c = 0; c < 4; ++c) {
        out.child[c] = rt::BVH4_EMPTY;
    }
    for (int c = 0; c < n_slots; ++c) {
        for (int a = 0; a < 3; ++a) {
            const float flo = floorf((slots[c].box.lo[a] - lo[a]) / scale[a]);
            const float fhi = ceilf((slots[c].box.hi[a] - lo[a]) / scale[a]);
            int qlo = int(fminf(fmaxf(flo, 0.0f), 255.0f)), qhi = int(fminf(fmaxf(fhi, 0.0f), 255.0f));
            // containment in REAL arithmetic (org + q * step is exact in double): what the error budget of the device's
            // parameter-space slab test starts from (rt_bvh4.h: bvh4_test_node)
            while (qlo > 0 && real_plane(uint32_t(qlo), scale[a], lo[a]) > double(slots[c].box.lo[a])) {
                --qlo;
            }
            while (qhi < 255 && real_plane(uint32_t(qhi), scale[a], lo[a]) < double(slots[c].box.hi[a])) {
                ++qhi;
            }
            if (real_plane(uint32_t(qlo), scale[a], lo[a]) > double(slots[c].box.lo[a]) ||
                real_plane(uint32_t(qhi), scale[a], lo[a]) < double(slots[c].box.hi[a])) {
                return false;
            }
            out.qlo[a] |= uint32_t(qlo) << (8 * c);
            out.qhi[a] |= uint32_t(qhi) << (8 * c);
        }
    }
    return true;
}

// the (up to four) children of the wide node that stands for BVH2 node `n`: its two children, then the child with the largest
// box is opened, twice at most (leaves stay)
RT_HD int wide_children(const rayhip_bvh2_node *nodes, const uint32_t n, Slot slots[4]) {
    children_of(nodes[n], slots);
    int n_slots = 2;
    while (n_slots < 4) {
        int best = -1;
        float best_area = -1.0f;
        for (int c = 0; c < n_slots; ++c) {
            if (!is_leaf(slots[c].ref) && half_area(slots[c].box) > best_area) {
                best_area = half_area(slots[c].box), best = c;
            }
        }
        if (best < 0) {
            break;
        }
        Slot two[2];
        children_of(nodes[slots[best].ref], two);
        slots[best] = two[0];
        slots[n_slots++] = two[1];
    }
    return n_slots;
}

struct Result {
    std::vector<rt::Bvh4Node> nodes;
    std::vector<uint32_t> blas_root4; // per mesh instance: root of its 4-wide BLAS (0xffffffff: not referenced)
    bool ok = false;
};

// The distinct bottom-level roots the top level references, in first-visit order, and for every mesh instance the ordinal of
// its root (0xffffffff: instance not referenced) -- the device driver (bvh4_build.hip.h) numbers the wide roots that way.
inline bool collect_roots(const rayhip_bvh2_node *nodes, const uint32_t n_nodes, const rayhip_mesh_instance *mis, const uint32_t n_mis,
                          const uint32_t tlas_root, std::vector<uint32_t> &roots, std::vector<uint32_t> &root_of_instance) {
    roots.clear();
    root_of_instance.assign(n_mis, 0xffffffffu);
    if (tlas_root == 0xffffffffu || tlas_root >= n_nodes) {
        return false;
    }
    std::vector<uint32_t> ordinal(n_nodes, 0xffffffffu);
    std::vector<uint32_t> stack = {tlas_root};
    size_t visited = 0;
    while (!stack.empty()) {
        const uint32_t n = stack.back();
        stack.pop_back();
        if (n >= n_nodes || ++visited > n_nodes) {
            return false;
        }
        const uint32_t ch[2] = {nodes[n].left_child, nodes[n].right_child};
        for (int k = 0; k < 2; ++k) {
            if (is_leaf(ch[k])) {
                const uint32_t mi = ch[k] & rt::BVH2_PRIM_INDEX_BITS;
                if (mi >= n_mis || mis[mi].node_index >= n_nodes) {
                    return false;
                }
                const uint32_t root2 = mis[mi].node_index;
                if (ordinal[root2] == 0xffffffffu) {
                    ordinal[root2] = uint32_t(roots.size());
                    roots.push_back(root2);
                }
                root_of_instance[mi] = ordinal[root2];
            } else {
                stack.push_back(ch[k]);
            }
        }
    }
    return true;
}

// nodes / mesh instances as they will be uploaded (i.e. after bvh_layout).  Only instances referenced by TLAS leaves
// are followed (the instance array is a sparse pool).
inline Result build(const rayhip_bvh2_node *nodes, const uint32_t n_nodes, const rayhip_mesh_instance *mis, const uint32_t n_mis,
                    const uint32_t tlas_root) {
    Result out;
    out.blas_root4.assign(n_mis, 0xffffffffu);
    if (tlas_root == 0xffffffffu || tlas_root >= n_nodes) {
        return out;
    }
    // instances referenced by the TLAS
    std::vector<uint32_t> inst;
    {
        std::vector<uint32_t> stack = {tlas_root};
        size_t visited = 0;
        while (!stack.empty()) {
            const uint32_t n = stack.back();
            stack.pop_back();
            if (n >= n_nodes || ++visited > n_nodes) {
                return out;
            }
            const uint32_t ch[2] = {nodes[n].left_child, nodes[n].right_child};
            for (int k = 0; k < 2; ++k) {
                if (is_leaf(ch[k])) {
                    const uint32_t mi = ch[k] & rt::BVH2_PRIM_INDEX_BITS;
                    if (mi >= n_mis) {
                        return out;
                    }
                    inst.push_back(mi);
                } else {
                    stack.push_back(ch[k]);
                }
            }
        }
    }
    std::vector<uint32_t> root4_of_bvh2(n_nodes, 0xffffffffu); // shared BLAS: build once
    struct Work {
        uint32_t bvh2_node, out_index;
    };
    for (const uint32_t mi : inst) {
        const uint32_t root2 = mis[mi].node_index;
        if (root2 >= n_nodes) {
            return out;
        }
        if (root4_of_bvh2[root2] != 0xffffffffu) {
            out.blas_root4[mi] = root4_of_bvh2[root2];
            continue;
        }
        const uint32_t root4 = uint32_t(out.nodes.size());
        out.nodes.emplace_back();
        std::vector<Work> stack = {Work{root2, root4}};
        size_t made = 0;
        while (!stack.empty()) {
            const Work w = stack.back();
            stack.pop_back();
            if (++made > size_t(n_nodes) + 1) {
                return out; // not a tree
            }
            Slot slots[4];
            const int n_slots = wide_children(nodes, w.bvh2_node, slots); // (links were bounds-checked by scene_validate.h)
            rt::Bvh4Node node;
            if (!quantise(slots, n_slots, node)) {
                return out;
            }
            // inner children get consecutive indices (one or two 128-byte lines), laid out before their subtrees
            const uint32_t first_child = uint32_t(out.nodes.size());
            uint32_t n_inner = 0;
            for (int c = 0; c < n_slots; ++c) {
                if (is_leaf(slots[c].ref)) {
                    node.child[c] = slots[c].ref;
                } else {
                    if (slots[c].ref >= n_nodes) {
                        return out;
                    }
                    node.child[c] = first_child + n_inner++;
                }
            }
            if (uint64_t(first_child) + n_inner >= rt::BVH4_SENTINEL) {
                return out;
            }
            out.nodes.resize(size_t(first_child) + n_inner);
            out.nodes[w.out_index] = node;
            for (int c = n_slots - 1; c >= 0; --c) { // first inner child is processed next (depth-first layout)
                if (!is_leaf(slots[c].ref)) {
                    stack.push_back(Work{slots[c].ref, node.child[c]});
                }
            }
        }
        root4_of_bvh2[root2] = root4;
        out.blas_root4[mi] = root4;
    }
    out.ok = true;
    return out;
}

} // namespace rayhip_bvh4
