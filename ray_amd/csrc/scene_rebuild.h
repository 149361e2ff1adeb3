// scene_rebuild.h -- both levels of the acceleration structure rebuilt with the linear builder of lbvh.h from what a flat
// scene holds: triangle records + vertices for the bottom level (one tree per mesh), instance transforms for the top level.
// Host driver (tests/hostsim, and the fallback of rayhip_scene_upload's validation); the device driver with the same
// element functions is lbvh.hip.h.  SURVEY.md section 8f, N1.
//
// What is reused and what is replaced: the triangle records (tri_accel_t, the precomputed planes every hit is computed
// from) are taken over entry by entry -- a hit's t / u / v and its triangle index cannot change -- while their ORDER, the
// leaf ranges and every node are new.  The reference pads leaves to 8 entries by repeating triangles
// (internal/Core.cpp:533-535); here every triangle of a mesh appears once (twice if it is alone in its leaf).
#pragma once

#include <string>
#include <unordered_map>
#include <vector>

#include "host_parallel.h"
#include "lbvh.h"

namespace rayhip_rebuild {

using rayhip_lbvh::Box;
constexpr uint32_t NONE = 0xffffffffu;
constexpr uint32_t COUNT_BITS = 7u << 29, INDEX_BITS = ~COUNT_BITS;

struct Rebuilt {
    std::vector<rayhip_bvh2_node> nodes;
    std::vector<rayhip_tri_accel> tris;
    std::vector<uint32_t> tri_indices;
    std::vector<rayhip_mesh_instance> mesh_instances;
    uint32_t tlas_root = NONE;
    bool ok = false;
    std::string why;
    // for the record
    uint32_t blas_nodes = 0, tlas_nodes = 0, unique_tris = 0;
};

// what the builder is fed: the bottom-level primitives (unique triangles per mesh) and the instances the top level holds
struct Gathered {
    std::vector<Box> prim_box;
    std::vector<uint32_t> prim_group; // dense mesh number
    std::vector<uint32_t> prim_entry; // the tris[] entry whose record represents the triangle
    std::vector<uint32_t> instances;  // mesh-instance slots the scene references
    std::vector<uint32_t> instance_group;
    uint32_t n_groups = 0;
};

inline Box triangle_box(const rayhip_scene_desc &d, const uint32_t tri) {
    Box b = rayhip_lbvh::empty_box();
    for (int k = 0; k < 3; ++k) {
        rayhip_lbvh::grow_point(b, d.vertices[d.vtx_indices[size_t(tri) * 3 + k]].p);
    }
    return b;
}

// leaf entry ranges of the tree under `root` (the host trees tell which tris[] entries a mesh owns when the scene carries
// no mesh table)
inline bool collect_leaf_ranges(const rayhip_bvh2_node *nodes, const uint32_t nodes_count, const uint32_t root,
                                std::vector<std::pair<uint32_t, uint32_t>> &ranges);
inline bool collect_leaf_ranges(const rayhip_scene_desc &d, const uint32_t root, std::vector<std::pair<uint32_t, uint32_t>> &ranges) {
    return collect_leaf_ranges(d.nodes, d.nodes_count, root, ranges);
}
inline bool collect_leaf_ranges(const rayhip_bvh2_node *nodes, const uint32_t nodes_count, const uint32_t root,
                                std::vector<std::pair<uint32_t, uint32_t>> &ranges) {
    std::vector<uint32_t> stack = {root};
    size_t visited = 0;
    while (!stack.empty()) {
        const uint32_t w = stack.back();
        stack.pop_back();
        if (w & COUNT_BITS) {
            ranges.emplace_back(w & INDEX_BITS, ((w & COUNT_BITS) >> 29) + 1);
            continue;
        }
        if (w >= nodes_count || ++visited > nodes_count) {
            return false;
        }
        stack.push_back(nodes[w].left_child);
        stack.push_back(nodes[w].right_child);
    }
    return true;
}

// the leaves of the host's top-level tree with the boxes they are stored with: (instance slot, box)
inline bool collect_leaf_boxes(const rayhip_scene_desc &d, const uint32_t root, std::vector<std::pair<uint32_t, Box>> &leaves) {
    std::vector<uint32_t> stack = {root};
    size_t visited = 0;
    while (!stack.empty()) {
        const uint32_t w = stack.back();
        stack.pop_back();
        if (w >= d.nodes_count || ++visited > d.nodes_count) {
            return false;
        }
        const rayhip_bvh2_node &n = d.nodes[w];
        for (int k = 0; k < 2; ++k) {
            const uint32_t link = k ? n.right_child : n.left_child;
            if (link & COUNT_BITS) {
                Box b;
                if (k == 0) {
                    b = Box{{n.ch_data0[0], n.ch_data0[2], n.ch_data2[0]}, {n.ch_data0[1], n.ch_data0[3], n.ch_data2[1]}};
                } else {
                    b = Box{{n.ch_data1[0], n.ch_data1[2], n.ch_data2[2]}, {n.ch_data1[1], n.ch_data1[3], n.ch_data2[3]}};
                }
                leaves.emplace_back(link & INDEX_BITS, b);
            } else {
                stack.push_back(link);
            }
        }
    }
    return true;
}

inline bool gather(const rayhip_scene_desc &d, Gathered &g, std::string &why) {
    if (d.tlas_root == NONE) {
        return true;
    }
    // instances: the leaves of the host's top level (the instance array is a sparse pool)
    {
        std::vector<std::pair<uint32_t, uint32_t>> leaves;
        if (!collect_leaf_ranges(d, d.tlas_root, leaves)) {
            why = "top-level tree is malformed";
            return false;
        }
        for (const auto &l : leaves) {
            if (l.first >= d.mesh_instances_count) {
                why = "top-level leaf names an instance outside the array";
                return false;
            }
            g.instances.push_back(l.first);
        }
        std::sort(g.instances.begin(), g.instances.end());
        g.instances.erase(std::unique(g.instances.begin(), g.instances.end()), g.instances.end());
    }
    // meshes -> dense groups; a mesh is identified by the root of its host tree (instances of one mesh share it)
    std::unordered_map<uint32_t, uint32_t> group_of_root;
    std::vector<uint32_t> roots;
    for (const uint32_t mi : g.instances) {
        const uint32_t root = d.mesh_instances[mi].node_index;
        auto it = group_of_root.find(root);
        if (it == group_of_root.end()) {
            it = group_of_root.emplace(root, uint32_t(roots.size())).first;
            roots.push_back(root);
        }
        g.instance_group.push_back(it->second);
    }
    g.n_groups = uint32_t(roots.size());
    // unique triangles per mesh: the entry with the lowest index represents a triangle
    std::vector<uint32_t> owner(d.vtx_indices_count / 3, NONE);
    for (uint32_t grp = 0; grp < g.n_groups; ++grp) {
        std::vector<std::pair<uint32_t, uint32_t>> ranges;
        if (!collect_leaf_ranges(d, roots[grp], ranges)) {
            why = "bottom-level tree is malformed";
            return false;
        }
        std::vector<uint32_t> tris_of_mesh;
        for (const auto &r : ranges) {
            for (uint32_t e = r.first; e < r.first + r.second; ++e) {
                if (e >= d.tris_count || d.tri_indices[e] >= owner.size()) {
                    why = "leaf entry outside the triangle arrays";
                    return false;
                }
                const uint32_t t = d.tri_indices[e];
                if (owner[t] == NONE) {
                    tris_of_mesh.push_back(t);
                }
                owner[t] = owner[t] < e ? owner[t] : e;
            }
        }
        std::sort(tris_of_mesh.begin(), tris_of_mesh.end()); // (deterministic input order for the stable sort of equal keys)
        for (const uint32_t t : tris_of_mesh) {
            g.prim_box.push_back(triangle_box(d, t));
            g.prim_group.push_back(grp);
            g.prim_entry.push_back(owner[t]);
        }
    }
    return true;
}

// world-space box of an object-space box under an instance transform (the eight corners)
inline Box transform_box(const Box &b, const float *xform) {
    Box o = rayhip_lbvh::empty_box();
    for (int k = 0; k < 8; ++k) {
        const rt::f3 p = rt::transform_point(rt::f3{(k & 1) ? b.hi[0] : b.lo[0], (k & 2) ? b.hi[1] : b.lo[1], (k & 4) ? b.hi[2] : b.lo[2]}, xform);
        const float q[3] = {p.x, p.y, p.z};
        rayhip_lbvh::grow_point(o, q);
    }
    return o;
}
inline Box node_box(const rayhip_bvh2_node &n) {
    Box b;
    b.lo[0] = fminf(n.ch_data0[0], n.ch_data1[0]), b.hi[0] = fmaxf(n.ch_data0[1], n.ch_data1[1]);
    b.lo[1] = fminf(n.ch_data0[2], n.ch_data1[2]), b.hi[1] = fmaxf(n.ch_data0[3], n.ch_data1[3]);
    b.lo[2] = fminf(n.ch_data2[0], n.ch_data2[2]), b.hi[2] = fmaxf(n.ch_data2[1], n.ch_data2[3]);
    return b;
}

// assemble the rebuilt scene arrays from the two builder outputs (shared by the host and the device driver)
inline void assemble(const rayhip_scene_desc &d, const Gathered &g, const rayhip_lbvh::Output &blas, const std::vector<Box> &instance_box,
                     const rayhip_lbvh::Output &tlas, Rebuilt &out) {
    (void)instance_box;
    out.nodes = blas.nodes;
    out.blas_nodes = uint32_t(blas.nodes.size());
    out.unique_tris = uint32_t(g.prim_entry.size());
    out.tris.resize(blas.entries.size());
    out.tri_indices.resize(blas.entries.size());
    for (size_t k = 0; k < blas.entries.size(); ++k) {
        const uint32_t e = g.prim_entry[blas.entries[k]];
        out.tris[k] = d.tris[e];
        out.tri_indices[k] = d.tri_indices[e];
    }
    out.mesh_instances.assign(d.mesh_instances, d.mesh_instances + d.mesh_instances_count);
    for (size_t k = 0; k < g.instances.size(); ++k) {
        out.mesh_instances[g.instances[k]].node_index = blas.group_root[g.instance_group[k]];
    }
    // top level: appended behind the bottom-level nodes; its inner links move by that offset, its leaves name instances
    const uint32_t base = uint32_t(out.nodes.size());
    out.tlas_nodes = uint32_t(tlas.nodes.size());
    for (rayhip_bvh2_node n : tlas.nodes) {
        for (uint32_t *link : {&n.left_child, &n.right_child}) {
            if ((*link & COUNT_BITS) == 0) {
                *link += base;
            } else {
                *link = (1u << 29) | g.instances[*link & INDEX_BITS];
            }
        }
        out.nodes.push_back(n);
    }
    out.tlas_root = tlas.group_root.empty() || tlas.group_root[0] == NONE ? NONE : base + tlas.group_root[0];
}

// `build`: the linear builder to run -- rayhip_lbvh::build_host, or the device driver of lbvh.hip.h (same element functions,
// same trees); signature  bool(const rayhip_lbvh::Input &, rayhip_lbvh::Output &, std::string &why)
struct HostBuilder {
    bool operator()(const rayhip_lbvh::Input &in, rayhip_lbvh::Output &out, std::string &) const {
        out = rayhip_lbvh::build_host(in);
        return true;
    }
};

// A builder for SMALL groups (leaf refinement: a group is one leaf of the scene's trees, at most 8 triangles): instead of
// the Morton order of the linear builder, every split is chosen by sweeping the surface-area heuristic over the three axes
// -- exhaustive for so few primitives.  Cost of a side: area x triangle tests, where a leaf of one triangle counts as two
// (the reference's leaf word cannot say "1": a lone triangle is stored, and tested, twice).  Same interface and output
// conventions as rayhip_lbvh::build_host; `leaf_is_primitive` / `roots_are_nodes` inputs are not supported.
// MEASURED against the linear builder as refinement builder (atrium, visit counters of the 4-wide walk over all bounces,
// leaf_max 2): 19.52 vs 19.66 node visits and 4.84 vs 5.16 triangle tests per ray -- the leaves of a regular mesh are small
// patches whose Morton order already is a good split order.  Not enough to move the refinement off the device builder; kept as
// the comparison (tests/hostsim: HOSTSIM_REFINE_SAH=1).
struct SmallSahBuilder {
    bool operator()(const rayhip_lbvh::Input &in, rayhip_lbvh::Output &out, std::string &why) const {
        using rayhip_lbvh::leaf_word;
        if (in.leaf_is_primitive || in.roots_are_nodes) {
            why = "SmallSahBuilder builds leaf refinements only";
            return false;
        }
        out = rayhip_lbvh::Output();
        out.group_root.assign(in.n_groups, NONE);
        out.bounds = rayhip_lbvh::empty_box();
        std::vector<std::vector<uint32_t>> members(in.n_groups);
        for (uint32_t p = 0; p < in.n_prims; ++p) {
            members[in.prim_group[p]].push_back(p);
        }
        auto half_area = [](const Box &b) {
            const float e[3] = {b.hi[0] - b.lo[0], b.hi[1] - b.lo[1], b.hi[2] - b.lo[2]};
            return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
        };
        auto tests = [&](const size_t m) { return m == 1 ? 2.0f : float(m); };
        struct Task {
            std::vector<uint32_t> ids;
            uint32_t node, side; // where the link goes (node == NONE: the group root)
            uint32_t group;
        };
        for (uint32_t grp = 0; grp < in.n_groups; ++grp) {
            if (members[grp].empty()) {
                continue;
            }
            if (members[grp].size() > 16) {
                why = "SmallSahBuilder: a group of " + std::to_string(members[grp].size()) + " primitives";
                return false;
            }
            std::vector<Task> stack;
            stack.push_back(Task{members[grp], NONE, 0, grp});
            while (!stack.empty()) {
                Task t = std::move(stack.back());
                stack.pop_back();
                Box box = rayhip_lbvh::empty_box();
                for (const uint32_t p : t.ids) {
                    rayhip_lbvh::grow(box, in.prim_box[p]);
                }
                uint32_t link;
                if (t.ids.size() <= in.leaf_max) {
                    const uint32_t first = uint32_t(out.entries.size());
                    for (const uint32_t p : t.ids) {
                        out.entries.push_back(p);
                    }
                    if (t.ids.size() == 1) {
                        out.entries.push_back(t.ids[0]);
                    }
                    link = leaf_word(first, uint32_t(t.ids.size()));
                } else {
                    // best (axis, position) by area x tests of the two sides
                    float best = 3.402823466e+38f;
                    int best_axis = 0;
                    size_t best_k = t.ids.size() / 2;
                    std::vector<uint32_t> sorted[3];
                    for (int a = 0; a < 3; ++a) {
                        sorted[a] = t.ids;
                        std::stable_sort(sorted[a].begin(), sorted[a].end(), [&](const uint32_t x, const uint32_t y) {
                            return in.prim_box[x].lo[a] + in.prim_box[x].hi[a] < in.prim_box[y].lo[a] + in.prim_box[y].hi[a];
                        });
                        const size_t n = sorted[a].size();
                        std::vector<Box> suffix(n + 1, rayhip_lbvh::empty_box());
                        for (size_t k = n; k-- > 0;) {
                            suffix[k] = suffix[k + 1];
                            rayhip_lbvh::grow(suffix[k], in.prim_box[sorted[a][k]]);
                        }
                        Box prefix = rayhip_lbvh::empty_box();
                        for (size_t k = 1; k < n; ++k) {
                            rayhip_lbvh::grow(prefix, in.prim_box[sorted[a][k - 1]]);
                            const float cost = half_area(prefix) * tests(k) + half_area(suffix[k]) * tests(n - k);
                            if (cost < best) {
                                best = cost, best_axis = a, best_k = k;
                            }
                        }
                    }
                    link = uint32_t(out.nodes.size());
                    out.nodes.emplace_back();
                    const std::vector<uint32_t> &srt = sorted[best_axis];
                    stack.push_back(Task{std::vector<uint32_t>(srt.begin() + best_k, srt.end()), link, 1, grp});
                    stack.push_back(Task{std::vector<uint32_t>(srt.begin(), srt.begin() + best_k), link, 0, grp});
                }
                if (t.node == NONE) {
                    out.group_root[grp] = link;
                } else {
                    rayhip_lbvh::write_child(out.nodes[t.node], int(t.side), box, link);
                }
                rayhip_lbvh::grow(out.bounds, box);
            }
        }
        return true;
    }
};

template <class Build> inline Rebuilt rebuild_with(const rayhip_scene_desc &d, const uint32_t leaf_max, Build &&build) {
    Rebuilt out;
    Gathered g;
    if (!gather(d, g, out.why)) {
        return out;
    }
    if (g.instances.empty()) {
        out.mesh_instances.assign(d.mesh_instances, d.mesh_instances + d.mesh_instances_count);
        out.ok = true;
        return out;
    }
    rayhip_lbvh::Input bi;
    bi.prim_box = g.prim_box.data(), bi.prim_group = g.prim_group.data(), bi.group_centroids = nullptr;
    bi.n_prims = uint32_t(g.prim_box.size()), bi.n_groups = g.n_groups, bi.leaf_max = leaf_max, bi.leaf_is_primitive = false, bi.roots_are_nodes = true;
    rayhip_lbvh::Output blas;
    if (!build(bi, blas, out.why)) {
        return out;
    }
    for (uint32_t grp = 0; grp < g.n_groups; ++grp) {
        if (blas.group_root[grp] == NONE) {
            out.why = "a mesh without triangles";
            return out;
        }
    }
    std::vector<Box> ibox;
    std::vector<uint32_t> igroup(g.instances.size(), 0);
    for (size_t k = 0; k < g.instances.size(); ++k) {
        ibox.push_back(transform_box(node_box(blas.nodes[blas.group_root[g.instance_group[k]]]), d.mesh_instances[g.instances[k]].xform));
    }
    rayhip_lbvh::Input ti;
    ti.prim_box = ibox.data(), ti.prim_group = igroup.data(), ti.group_centroids = nullptr;
    ti.n_prims = uint32_t(ibox.size()), ti.n_groups = 1, ti.leaf_max = 1, ti.leaf_is_primitive = true, ti.roots_are_nodes = true;
    rayhip_lbvh::Output tlas;
    if (!build(ti, tlas, out.why)) {
        return out;
    }
    assemble(d, g, blas, ibox, tlas, out);
    out.ok = true;
    return out;
}
inline Rebuilt rebuild_host(const rayhip_scene_desc &d, const uint32_t leaf_max) { return rebuild_with(d, leaf_max, HostBuilder()); }

// ---- leaf refinement -------------------------------------------------------------------------------------------------------------
// The reference's surface-area-heuristic trees are good trees with coarse leaves: up to 8 triangle slots each
// (bvh_settings_t::min_primitives_in_leaf = 8, SceneCPU.cpp:347), padded with repeated triangles -- a CPU SIMD layout.  On
// the GPU a triangle test costs about as much wave time as a node visit (measured: 116 vs 230 instructions at 30 % vs 48 % of
// the lanes busy), so a ray pays more for the ~15 triangle tests of a walk than for its last tree levels.  Refinement keeps
// the tree and replaces every leaf that holds more than `leaf_max` triangles by a small subtree over them (the linear
// builder, one group per leaf) -- same triangle records, same hits.
template <class Build> inline Rebuilt refine_with(const rayhip_scene_desc &d, const uint32_t leaf_max, Build &&build) {
    Rebuilt out;
    Gathered g;
    if (d.tlas_root == NONE) {
        out.nodes.assign(d.nodes, d.nodes + d.nodes_count);
        out.tris.assign(d.tris, d.tris + d.tris_count);
        out.tri_indices.assign(d.tri_indices, d.tri_indices + d.tri_indices_count);
        out.mesh_instances.assign(d.mesh_instances, d.mesh_instances + d.mesh_instances_count);
        out.tlas_root = d.tlas_root;
        out.ok = true;
        return out;
    }
    // the leaves of every referenced bottom-level tree, each one a group
    std::vector<std::pair<uint32_t, uint32_t>> top;
    if (!collect_leaf_ranges(d, d.tlas_root, top)) {
        out.why = "top-level tree is malformed";
        return out;
    }
    struct LeafSite {
        uint32_t node, side; // where the leaf word sits
    };
    std::vector<LeafSite> sites;
    std::vector<uint8_t> seen(d.nodes_count, 0);
    for (const auto &t : top) {
        if (t.first >= d.mesh_instances_count) {
            out.why = "top-level leaf names an instance outside the array";
            return out;
        }
        std::vector<uint32_t> stack = {d.mesh_instances[t.first].node_index};
        while (!stack.empty()) {
            const uint32_t w = stack.back();
            stack.pop_back();
            if (w >= d.nodes_count) {
                out.why = "bottom-level tree is malformed";
                return out;
            }
            if (seen[w]) {
                continue;
            }
            seen[w] = 1;
            const uint32_t ch[2] = {d.nodes[w].left_child, d.nodes[w].right_child};
            for (uint32_t k = 0; k < 2; ++k) {
                if (ch[k] & COUNT_BITS) {
                    sites.push_back(LeafSite{w, k});
                } else {
                    stack.push_back(ch[k]);
                }
            }
        }
    }
    g.n_groups = uint32_t(sites.size());
    // the unique triangles of every leaf (padding repeats triangles inside a leaf): counted, then written, both passes shared
    // out over the host cores (3.9 M entries of the Bistro-class scene: the boxes alone are 9 M vertex gathers)
    std::vector<uint32_t> first_prim(size_t(g.n_groups) + 1, 0);
    std::vector<uint8_t> bad(g.n_groups, 0);
    auto unique_entries = [&](const uint32_t grp, uint32_t uniq[8]) -> uint32_t {
        const uint32_t w = sites[grp].side ? d.nodes[sites[grp].node].right_child : d.nodes[sites[grp].node].left_child;
        const uint32_t first = w & INDEX_BITS, count = ((w & COUNT_BITS) >> 29) + 1;
        if (first + count > d.tris_count) {
            return 0xffffffffu;
        }
        uint32_t n_uniq = 0;
        for (uint32_t e = first; e < first + count; ++e) {
            bool dup = false;
            for (uint32_t k = 0; k < n_uniq; ++k) {
                dup |= d.tri_indices[uniq[k]] == d.tri_indices[e];
            }
            if (!dup) {
                uniq[n_uniq++] = e;
            }
        }
        return n_uniq;
    };
    rayhip_host::parallel_blocks(g.n_groups, 4096, [&](const size_t b, const size_t e) {
        for (size_t grp = b; grp < e; ++grp) {
            uint32_t uniq[8];
            const uint32_t n_uniq = unique_entries(uint32_t(grp), uniq);
            if (n_uniq == 0xffffffffu) {
                bad[grp] = 1;
                continue;
            }
            for (uint32_t k = 0; k < n_uniq; ++k) {
                bad[grp] |= d.tri_indices[uniq[k]] >= d.vtx_indices_count / 3 ? 2 : 0;
            }
            first_prim[grp + 1] = n_uniq;
        }
    });
    for (uint32_t grp = 0; grp < g.n_groups; ++grp) {
        if (bad[grp]) {
            out.why = (bad[grp] & 1) ? "leaf range outside the triangle array" : "leaf entry outside the triangle arrays";
            return out;
        }
        first_prim[grp + 1] += first_prim[grp];
    }
    const size_t n_prims_total = first_prim[g.n_groups];
    g.prim_box.resize(n_prims_total), g.prim_group.resize(n_prims_total), g.prim_entry.resize(n_prims_total);
    rayhip_host::parallel_blocks(g.n_groups, 4096, [&](const size_t b, const size_t e) {
        for (size_t grp = b; grp < e; ++grp) {
            uint32_t uniq[8];
            const uint32_t n_uniq = unique_entries(uint32_t(grp), uniq);
            for (uint32_t k = 0; k < n_uniq; ++k) {
                const size_t at = size_t(first_prim[grp]) + k;
                g.prim_box[at] = triangle_box(d, d.tri_indices[uniq[k]]);
                g.prim_group[at] = uint32_t(grp);
                g.prim_entry[at] = uniq[k];
            }
        }
    });
    rayhip_lbvh::Input bi;
    bi.prim_box = g.prim_box.data(), bi.prim_group = g.prim_group.data(), bi.group_centroids = nullptr;
    bi.n_prims = uint32_t(g.prim_box.size()), bi.n_groups = g.n_groups, bi.leaf_max = leaf_max, bi.leaf_is_primitive = false,
    bi.roots_are_nodes = false;
    rayhip_lbvh::Output sub;
    if (!build(bi, sub, out.why)) {
        return out;
    }
    // splice: the old nodes keep their indices, the subtrees follow them
    const uint32_t base = d.nodes_count;
    out.nodes.resize(size_t(d.nodes_count) + sub.nodes.size());
    rayhip_host::parallel_blocks(d.nodes_count, 1 << 16, [&](const size_t b, const size_t e) {
        memcpy(static_cast<void *>(out.nodes.data() + b), static_cast<const void *>(d.nodes + b), (e - b) * sizeof(rayhip_bvh2_node));
    });
    rayhip_host::parallel_blocks(sub.nodes.size(), 1 << 16, [&](const size_t b, const size_t e) {
        for (size_t i = b; i < e; ++i) {
            rayhip_bvh2_node n = sub.nodes[i];
            for (uint32_t *link : {&n.left_child, &n.right_child}) {
                if ((*link & COUNT_BITS) == 0) {
                    *link += base;
                }
            }
            out.nodes[size_t(base) + i] = n;
        }
    });
    for (uint32_t grp = 0; grp < g.n_groups; ++grp) {
        uint32_t link = sub.group_root[grp];
        if (link == NONE) {
            out.why = "an empty leaf";
            return out;
        }
        if ((link & COUNT_BITS) == 0) {
            link += base;
        }
        (sites[grp].side ? out.nodes[sites[grp].node].right_child : out.nodes[sites[grp].node].left_child) = link;
    }
    out.tris.resize(sub.entries.size()), out.tri_indices.resize(sub.entries.size());
    rayhip_host::parallel_blocks(sub.entries.size(), 1 << 16, [&](const size_t b, const size_t e2) {
        for (size_t k = b; k < e2; ++k) {
            const uint32_t e = g.prim_entry[sub.entries[k]];
            out.tris[k] = d.tris[e], out.tri_indices[k] = d.tri_indices[e];
        }
    });
    out.mesh_instances.assign(d.mesh_instances, d.mesh_instances + d.mesh_instances_count);
    out.tlas_root = d.tlas_root;
    out.blas_nodes = uint32_t(sub.nodes.size()), out.unique_tris = uint32_t(g.prim_entry.size());
    out.ok = true;
    return out;
}
inline Rebuilt refine_host(const rayhip_scene_desc &d, const uint32_t leaf_max) { return refine_with(d, leaf_max, HostBuilder()); }
inline Rebuilt refine_host_sah(const rayhip_scene_desc &d, const uint32_t leaf_max) { return refine_with(d, leaf_max, SmallSahBuilder()); }

} // namespace rayhip_rebuild
