// rt_arealights.h -- K4: implicit hits of analytic lights (sphere, directional, rect, disk, line, env) through the
// 8-wide quantised light tree.
//
// Restates
//   IntersectAreaLights(rays, lights, light_cwbvh nodes, inout hits)   reference internal/CoreRef.cpp:3616-3860
//       secondary rays only (TraceRays(..., trace_lights = true), RendererCPU.h:537-539; CoreRef.cpp:4847-4849).
//       A light that is hit closer than the surface hit replaces it: obj_index = -light-1, v = 0, t, and
//       u = the probability with which NEE would have picked this light from the ray origin (product of the
//       normalised importances along the tree path) -- the shading kernel turns it into the MIS weight.
//   IntersectAreaLights(shadow_ray_t, ...) -> 0/1                       CoreRef.cpp:4451-4592
//       rect / disk "blocker" lights occlude shadow rays.
//   bbox_test_oct(cwbvh_node_t)   CoreRef.cpp:393-477      TraversalStack::sort_top3/4/N   CoreRef.cpp:508-590
//   quadratic                     CoreRef.cpp:750-764
//
// The per-ray stack lives in LDS in the kernels (LightStackLds, round 3; a private array before: scratch memory) and in a plain
// array on the host build; this stage only runs when the scene has visible / blocker analytic lights.
#pragma once

#include "shade_lights.h"
#include "rt_traverse.h"

namespace rt {

struct LightStackEntry {
    uint32_t index;
    float dist;
    float factor;
};

// Where the entries live.  Private: an array of the caller (the host build; scratch memory if a kernel used it).  Lds: three
// depth-major planes of the wavefront's shared memory, plane[depth][lane] -- bank == lane at any mix of depths, conflict-free,
// the layout of the traversal stacks (kernels.hip.h); the reference's shader keeps the same 48 entries per invocation in
// shared memory (shaders/intersect_area_lights.comp.glsl:52-53).
struct LightStackPrivate {
    LightStackEntry e[MAX_STACK_SIZE];
    RT_HD LightStackEntry get(const uint32_t i) const { return e[i]; }
    RT_HD void set(const uint32_t i, const LightStackEntry &v) { e[i] = v; }
    RT_HD float dist(const uint32_t i) const { return e[i].dist; }
};
constexpr int LIGHT_STACK_LDS_WORDS = 3 * MAX_STACK_SIZE * 64; // per wavefront
struct LightStackLds {
    uint32_t *lane_base; // &planes[0][0][lane]
    RT_HD LightStackEntry get(const uint32_t i) const {
        return LightStackEntry{lane_base[i * 64u], uint_as_float(lane_base[(MAX_STACK_SIZE + i) * 64u]), uint_as_float(lane_base[(2 * MAX_STACK_SIZE + i) * 64u])};
    }
    RT_HD void set(const uint32_t i, const LightStackEntry &v) {
        lane_base[i * 64u] = v.index, lane_base[(MAX_STACK_SIZE + i) * 64u] = float_as_uint(v.dist), lane_base[(2 * MAX_STACK_SIZE + i) * 64u] = float_as_uint(v.factor);
    }
    RT_HD float dist(const uint32_t i) const { return uint_as_float(lane_base[(MAX_STACK_SIZE + i) * 64u]); }
};

template <class Store> struct LightStackT {
    Store s;
    uint32_t size = 0;
    RT_HD void push(uint32_t index, float dist, float factor) {
        if (size < uint32_t(MAX_STACK_SIZE)) { // (the reference asserts)
            s.set(size, LightStackEntry{index, dist, factor});
        }
        ++size;
    }
    RT_HD LightStackEntry pop() {
        --size;
        return size < uint32_t(MAX_STACK_SIZE) ? s.get(size) : LightStackEntry{0u, 3.402823466e+38f, 0.0f}; // overflowed entries are dropped
    }
    RT_HD void swap(uint32_t a, uint32_t b) {
        const LightStackEntry t = s.get(a);
        s.set(a, s.get(b));
        s.set(b, t);
    }
    // Ordering the topmost entries by descending distance (the nearest is popped first).  The reference has three routines for
    // 3, 4 and N entries (CoreRef.cpp:508-590) and they are NOT one sort: on equal distances they leave different permutations
    // (checked exhaustively over all tie patterns: 18 of 27 triples and 56 of 256 quadruples come out differently from a stable
    // insertion), and which light a tie pops first is visible in the result.  Each is therefore restated with ITS decisions:
    //   3 entries: four strict comparisons select one of six orders (a table of the decision tree's leaves);
    //   4 entries: the five strict compare-exchanges (0,1) (2,3) (0,2) (1,3) (1,2), as a loop over the pair list;
    //   N entries: insertion from the left, an entry moves up past strictly smaller ones only.
    RT_HD void order_top(const int count) {
        const uint32_t base = size - uint32_t(count);
        if (count == 3) {
            const LightStackEntry e[3] = {s.get(base), s.get(base + 1), s.get(base + 2)};
            const bool ab = e[0].dist > e[1].dist, bc = e[1].dist > e[2].dist, ac = e[0].dist > e[2].dist, cb = e[2].dist > e[1].dist;
            // leaves of the tree, as "which entry goes to slot 0 / 1 / 2", two bits each
            const uint32_t abc = 0u | (1u << 2) | (2u << 4), acb = 0u | (2u << 2) | (1u << 4), cab = 2u | (0u << 2) | (1u << 4);
            const uint32_t bac = 1u | (0u << 2) | (2u << 4), cba = 2u | (1u << 2) | (0u << 4), bca = 1u | (2u << 2) | (0u << 4);
            const uint32_t order = ab ? (bc ? abc : (ac ? acb : cab)) : (ac ? bac : (cb ? cba : bca));
            if (order != abc) {
                for (uint32_t k = 0; k < 3; ++k) {
                    s.set(base + k, e[(order >> (2 * k)) & 3u]);
                }
            }
        } else if (count == 4) {
            const uint32_t pairs[5][2] = {{0, 1}, {2, 3}, {0, 2}, {1, 3}, {1, 2}};
            for (const auto &pr : pairs) {
                if (s.dist(base + pr[0]) < s.dist(base + pr[1])) {
                    swap(base + pr[0], base + pr[1]);
                }
            }
        } else {
            for (uint32_t i = base + 1; i < size; ++i) {
                const LightStackEntry moving = s.get(i);
                uint32_t slot = i;
                while (slot > base && s.dist(slot - 1) < moving.dist) {
                    s.set(slot, s.get(slot - 1));
                    --slot;
                }
                s.set(slot, moving);
            }
        }
    }
};
typedef LightStackT<LightStackPrivate> LightStack;

// slab test of the eight quantised child boxes; bit i of the result = child i is hit within [0, t]
RT_HD uint32_t bbox_test_oct(const f3 o, const f3 inv_d, const float t, const rayhip_light_cwbvh_node &n, float out_dist[8]) {
    uint32_t mask = 0;
    for (int i = 0; i < 8; ++i) {
        float bmin[3], bmax[3];
        light_child_box(n, i, bmin, bmax);
        if (bbox_test(o, inv_d, t, bmin, bmax, out_dist[i])) {
            mask |= (1u << i);
        }
    }
    return mask;
}

RT_HD int first_bit(const uint32_t mask) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffs(int(mask)) - 1;
#else
    return __builtin_ctz(mask);
#endif
}

// CoreRef.cpp:750-764
RT_HD bool quadratic(const float a, const float b, const float c, float &t0, float &t1) {
    const float d = b * b - 4.0f * a * c;
    if (d < 0.0f) {
        return false;
    }
    const float sqrt_d = sqrtf(d);
    float q;
    if (b < 0.0f) {
        q = -0.5f * (b - sqrt_d);
    } else {
        q = -0.5f * (b + sqrt_d);
    }
    t0 = q / a;
    t1 = c / q;
    return true;
}

// Push the children selected by `mask` the way all of the reference's 8-wide walks do (1 hit: descend; 2: nearer
// first; 3, 4, 5+: push all, sort the top, pop).  Returns false when no child was hit.
template <class Stack>
RT_HD bool light_tree_descend(const rayhip_light_cwbvh_node &n, uint32_t mask, const float dist[8], const float factors[8],
                              Stack &st, LightStackEntry &cur) {
    if (mask == 0) {
        return false;
    }
    int i = first_bit(mask);
    mask &= mask - 1;
    if (mask == 0) { // one box
        cur.index = n.child[i];
        cur.factor *= factors[i];
        return true;
    }
    const int i2 = first_bit(mask);
    mask &= mask - 1;
    if (mask == 0) { // two boxes
        if (dist[i] < dist[i2]) {
            st.push(n.child[i2], dist[i2], cur.factor * factors[i2]);
            cur.index = n.child[i];
            cur.factor *= factors[i];
        } else {
            st.push(n.child[i], dist[i], cur.factor * factors[i]);
            cur.index = n.child[i2];
            cur.factor *= factors[i2];
        }
        return true;
    }
    st.push(n.child[i], dist[i], cur.factor * factors[i]);
    st.push(n.child[i2], dist[i2], cur.factor * factors[i2]);

    i = first_bit(mask);
    mask &= mask - 1;
    st.push(n.child[i], dist[i], cur.factor * factors[i]);
    if (mask == 0) { // three
        st.order_top(3);
        cur = st.pop();
        return true;
    }
    i = first_bit(mask);
    mask &= mask - 1;
    st.push(n.child[i], dist[i], cur.factor * factors[i]);
    if (mask == 0) { // four
        st.order_top(4);
        cur = st.pop();
        return true;
    }
    const uint32_t size_before = st.size;
    do { // five to eight
        i = first_bit(mask);
        mask &= mask - 1;
        st.push(n.child[i], dist[i], cur.factor * factors[i]);
    } while (mask != 0);
    st.order_top(int(st.size - size_before + 4));
    cur = st.pop();
    return true;
}

// CoreRef.cpp:3616-3860
// (`st`: an empty stack -- LightStack on the host, LightStackT<LightStackLds> in the kernel)
template <class Stack>
RT_HD void intersect_area_lights(const SceneView &sc, const f3 ro, const f3 rd, const uint32_t ray_depth, Hit &inter, Stack &st) {
    const uint32_t ray_flags = (1u << get_ray_type(ray_depth));
    const f3 inv_d = safe_invert(rd);

    st.size = 0;
    st.push(0u, 0.0f, 1.0f);

    while (st.size != 0) {
        LightStackEntry cur = st.pop();
        if (cur.dist > inter.t || cur.factor == 0.0f) {
            continue;
        }
        // the reference's `goto TRAVERSE` loop: walk down while inner nodes keep being hit
        bool reached_leaf = true;
        while ((cur.index & LEAF_NODE_BIT) == 0) {
            const rayhip_light_cwbvh_node &n = sc.light_cwnodes[cur.index];
            float dist[8];
            const uint32_t mask = bbox_test_oct(ro, inv_d, inter.t, n, dist);
            if (mask == 0) {
                reached_leaf = false;
                break;
            }
            float factors[8];
            light_node_importances(sc, cur.index, ro, factors);
            const float total_importance = sum8_sse_order(factors);
            if (total_importance == 0.0f) {
                reached_leaf = false;
                break;
            }
            for (int k = 0; k < 8; ++k) {
                factors[k] /= total_importance;
            }
            light_tree_descend(n, mask, dist, factors, st, cur);
        }
        if (!reached_leaf) {
            continue;
        }

        const int light_index = int(cur.index & PRIM_INDEX_BITS);
        const rayhip_light &l = sc.lights[light_index];
        if (!light_visible(l) || (light_ray_visibility(l) & ray_flags) == 0) {
            continue;
        }
        if (light_sky_portal(l) && inter.v >= 0.0f) {
            continue; // portal lights affect only missed rays
        }
        const bool no_shadow = !light_cast_shadow(l);
        const uint32_t ltype = light_type(l);
        if (ltype == LIGHT_TYPE_SPHERE) {
            const f3 light_pos = mk3(&l.params[0]);
            const float radius = l.params[7], spot = l.params[8];
            const f3 op = light_pos - ro;
            const float b = dot(op, rd);
            float det = b * b - dot(op, op) + radius * radius;
            if (det >= 0.0f) {
                det = sqrtf(det);
                const float t1 = b - det, t2 = b + det;
                if (t1 > HIT_EPS && (t1 < inter.t || no_shadow)) {
                    bool accept = true;
                    if (spot > 0.0f) {
                        const float _dot = -dot(rd, mk3(&l.params[4]));
                        if (_dot > 0.0f) {
                            const float _angle = acosf(saturatef(_dot));
                            accept &= (_angle <= spot);
                        } else {
                            accept = false;
                        }
                    }
                    if (accept) {
                        inter.v = 0.0f;
                        inter.obj_index = -light_index - 1;
                        inter.t = t1;
                        inter.u = cur.factor;
                    }
                } else if (t2 > HIT_EPS && (t2 < inter.t || no_shadow)) {
                    inter.v = 0.0f;
                    inter.obj_index = -light_index - 1;
                    inter.t = t2;
                    inter.u = cur.factor;
                }
            }
        } else if (ltype == LIGHT_TYPE_DIR) {
            const f3 light_dir = mk3(&l.params[0]);
            const float cos_angle = l.params[3];
            const float cos_theta = dot(rd, light_dir);
            if ((inter.v < 0.0f || no_shadow) && cos_theta > cos_angle) {
                inter.v = 0.0f;
                inter.obj_index = -light_index - 1;
                inter.t = 1.0f / cos_theta;
                inter.u = cur.factor;
            }
        } else if (ltype == LIGHT_TYPE_RECT || ltype == LIGHT_TYPE_DISK) {
            const f3 light_pos = mk3(&l.params[0]);
            f3 light_u = mk3(&l.params[4]), light_v = mk3(&l.params[8]);
            const f3 light_forward = normalize(cross(light_u, light_v));

            const float plane_dist = dot(light_forward, light_pos);
            const float cos_theta = dot(rd, light_forward);
            // (rect spells it `/ fminf(cos_theta, -FLT_EPS)`, disk `safe_div_neg`: the same expression)
            const float t = safe_div_neg(plane_dist - dot(light_forward, ro), cos_theta);

            if (cos_theta < 0.0f && t > HIT_EPS && (t < inter.t || no_shadow)) {
                light_u = light_u / dot(light_u, light_u);
                light_v = light_v / dot(light_v, light_v);

                const f3 p = ro + rd * t;
                const f3 vi = p - light_pos;
                const float a1 = dot(light_u, vi);
                bool hit;
                if (ltype == LIGHT_TYPE_RECT) {
                    hit = false;
                    if (a1 >= -0.5f && a1 <= 0.5f) {
                        const float a2 = dot(light_v, vi);
                        hit = (a2 >= -0.5f && a2 <= 0.5f);
                    }
                } else {
                    const float a2 = dot(light_v, vi);
                    hit = sqrtf(a1 * a1 + a2 * a2) <= 0.5f;
                }
                if (hit) {
                    inter.v = 0.0f;
                    inter.obj_index = -light_index - 1;
                    inter.t = t;
                    inter.u = cur.factor;
                }
            }
        } else if (ltype == LIGHT_TYPE_LINE) {
            const f3 light_pos = mk3(&l.params[0]);
            const f3 light_u = mk3(&l.params[4]), light_dir = mk3(&l.params[8]);
            const float line_radius = l.params[7], line_height = l.params[11];
            const f3 light_v = cross(light_u, light_dir);

            const f3 ro0 = ro - light_pos;
            const f3 _ro = {dot(ro0, light_dir), dot(ro0, light_u), dot(ro0, light_v)};
            const f3 _rd = {dot(rd, light_dir), dot(rd, light_u), dot(rd, light_v)};

            const float A = _rd.z * _rd.z + _rd.y * _rd.y;
            const float B = 2.0f * (_rd.z * _ro.z + _rd.y * _ro.y);
            const float C = _ro.z * _ro.z + _ro.y * _ro.y - line_radius * line_radius;

            float t0, t1;
            if (quadratic(A, B, C, t0, t1) && t0 > HIT_EPS && t1 > HIT_EPS) {
                const float t = fminf(t0, t1);
                const f3 p = _ro + t * _rd;
                if (fabsf(p.x) < 0.5f * line_height && (t < inter.t || no_shadow)) {
                    inter.v = 0.0f;
                    inter.obj_index = -light_index - 1;
                    inter.t = t;
                    inter.u = cur.factor;
                }
            }
        } else if (ltype == LIGHT_TYPE_ENV && inter.v < 0.0f) {
            // NOTE: mask remains empty
            inter.obj_index = -light_index - 1;
            inter.u = cur.factor;
        }
    }
}

// CoreRef.cpp:4451-4592: 0 if a rect / disk blocker light lies between the shadow ray's ends, else 1
template <class Stack>
RT_HD float intersect_area_lights_shadow(const SceneView &sc, const ShadowRay &r, Stack &st) {
    const float rdist = fabsf(r.dist);
    const f3 ro = r.o, rd = r.d;
    const f3 inv_d = safe_invert(rd);
    const float ones[8] = {1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.0f};

    st.size = 0;
    st.push(0u, 0.0f, 1.0f);

    while (st.size != 0) {
        LightStackEntry cur = st.pop();
        if (cur.dist > rdist) {
            continue;
        }
        bool reached_leaf = true;
        while ((cur.index & LEAF_NODE_BIT) == 0) {
            const rayhip_light_cwbvh_node &n = sc.light_cwnodes[cur.index];
            float dist[8];
            const uint32_t mask = bbox_test_oct(ro, inv_d, rdist, n, dist);
            if (!light_tree_descend(n, mask, dist, ones, st, cur)) {
                reached_leaf = false;
                break;
            }
        }
        if (!reached_leaf) {
            continue;
        }
        const int light_index = int(cur.index & PRIM_INDEX_BITS);
        const rayhip_light &l = sc.lights[light_index];
        if ((light_ray_visibility(l) & (1u << RAY_TYPE_SHADOW)) == 0) {
            continue;
        }
        if (light_sky_portal(l) && r.dist >= 0.0f) {
            continue;
        }
        const uint32_t ltype = light_type(l);
        if (ltype == LIGHT_TYPE_RECT || ltype == LIGHT_TYPE_DISK) {
            const f3 light_pos = mk3(&l.params[0]);
            f3 light_u = mk3(&l.params[4]), light_v = mk3(&l.params[8]);
            const f3 light_forward = normalize(cross(light_u, light_v));

            const float plane_dist = dot(light_forward, light_pos);
            const float cos_theta = dot(rd, light_forward);
            const float t = safe_div_neg(plane_dist - dot(light_forward, ro), cos_theta);

            if (cos_theta < 0.0f && t > HIT_EPS && t < rdist) {
                light_u = light_u / dot(light_u, light_u);
                light_v = light_v / dot(light_v, light_v);

                const f3 p = ro + rd * t;
                const f3 vi = p - light_pos;
                const float a1 = dot(light_u, vi);
                if (ltype == LIGHT_TYPE_RECT) {
                    if (a1 >= -0.5f && a1 <= 0.5f) {
                        const float a2 = dot(light_v, vi);
                        if (a2 >= -0.5f && a2 <= 0.5f) {
                            return 0.0f;
                        }
                    }
                } else {
                    const float a2 = dot(light_v, vi);
                    if (sqrtf(a1 * a1 + a2 * a2) <= 0.5f) {
                        return 0.0f;
                    }
                }
            }
        }
    }
    return 1.0f;
}

} // namespace rt
