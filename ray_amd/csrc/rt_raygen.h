// rt_raygen.h -- primary ray generation.  Restates Ref::GeneratePrimaryRays, reference
// internal/CoreRef.cpp:1429-1553 (GLSL twin: shaders/primary_ray_gen.comp.glsl:71-164).
#pragma once

#include "rt_rng.h"
#include "rt_types.h"

namespace rt {

// Per-frame constants; the libm calls on camera-only terms (tanf/atanf, CoreRef.cpp:1439-1442) are made
// once on the host by the C-ABI layer, so they are bit-identical to the reference.
struct RayGenParams {
    f3 origin, fwd, side, up;
    float focus_distance;
    float k;            // float(w) / float(h)
    float fov_k;        // tanf(0.5*fov*PI/180) * focus_distance
    float spread_angle; // atanf(2*tan / h)
    float shift[2];
    float fstop, focal_length, sensor_height, lens_rotation, lens_ratio;
    int lens_blades;
    float clip_start, clip_end;
    int filter_is_box;
    int w, h;
    int rect[4];
    int iteration;
    uint32_t rand_seed;
    Shard shard;
    int skip_ior; // 1: the ior plane of the rays is not written (ShadeParams::plain_ior: nobody will read it)
};

// CoreRef.cpp:1452-1467
RT_HD float lookup_filter_table(const float *filter_table, float x) {
    x *= (FILTER_TABLE_SIZE - 1);
    const int index = int(x) < (FILTER_TABLE_SIZE - 1) ? int(x) : (FILTER_TABLE_SIZE - 1);
    const int nindex = (index + 1) < (FILTER_TABLE_SIZE - 1) ? (index + 1) : (FILTER_TABLE_SIZE - 1);
    const float t = x - float(index);
    const float data0 = filter_table[index];
    if (t == 0.0f) {
        return data0;
    }
    const float data1 = filter_table[nindex];
    return (1.0f - t) * data0 + t * data1;
}

// CoreRef.cpp:767-769
RT_HD float ngon_rad(const float theta, const float n) {
    return portable_cos(PI / n) / portable_cos(theta - (2.0f * PI / n) * floorf((n * theta + PI) / (2.0f * PI)));
}

// one pixel (x, y) -> camera ray + initial hit record (clip range in hit.t)
RT_HD void generate_primary_ray(const RayGenParams &p, const uint32_t *pmj, const float *filter_table, const int x,
                                const int y, Ray &out_r, Hit &out_i) {
    float fx = float(x), fy = float(y);

    const uint32_t px_hash = hash(uint32_t((x << 16) | y));
    const uint32_t rand_hash = hash_combine(px_hash, p.rand_seed);

    const f2 filter_rand = get_scrambled_2d_rand(RAND_DIM_FILTER, rand_hash, p.iteration - 1, pmj);
    float rx = filter_rand.x, ry = filter_rand.y;
    if (!p.filter_is_box) {
        rx = lookup_filter_table(filter_table, rx);
        ry = lookup_filter_table(filter_table, ry);
    }
    fx += rx;
    fy += ry;

    f2 offset = {0.0f, 0.0f};
    if (p.fstop > 0.0f) {
        const f2 lens_rand = get_scrambled_2d_rand(RAND_DIM_LENS, rand_hash, p.iteration - 1, pmj);
        offset = {2.0f * lens_rand.x - 1.0f, 2.0f * lens_rand.y - 1.0f};
        if (offset.x != 0.0f && offset.y != 0.0f) {
            float theta, r;
            if (fabsf(offset.x) > fabsf(offset.y)) {
                r = offset.x;
                theta = 0.25f * PI * (offset.y / offset.x);
            } else {
                r = offset.y;
                theta = 0.5f * PI - 0.25f * PI * (offset.x / offset.y);
            }
            if (p.lens_blades) {
                r *= ngon_rad(theta, float(p.lens_blades));
            }
            theta += p.lens_rotation;

            const f2 sincos_theta = portable_sincos(theta);
            offset.x = 0.5f * r * sincos_theta.y / p.lens_ratio;
            offset.y = 0.5f * r * sincos_theta.x;
        }
        const float coc = 0.5f * (p.focal_length / p.fstop);
        offset = offset * (coc * p.sensor_height);
    }

    const f3 _origin = p.origin + p.side * offset.x + p.up * offset.y;

    // get_pix_dir, CoreRef.cpp:1445-1450
    f3 pp = {2 * p.fov_k * (fx / float(p.w) + p.shift[0] / p.k) - p.fov_k,
             2 * p.fov_k * (-fy / float(p.h) + p.shift[1]) + p.fov_k, p.focus_distance};
    pp = p.origin + p.k * pp.x * p.side + pp.y * p.up + pp.z * p.fwd;
    const f3 _d = normalize(pp - _origin);

    const float clip_start = p.clip_start / dot(_d, p.fwd);

    out_r.o = _origin + _d * clip_start;
    out_r.d = _d;
    out_r.c = {1.0f, 1.0f, 1.0f};
    out_r.ior[0] = out_r.ior[1] = out_r.ior[2] = out_r.ior[3] = -1.0f; // air ior is implicit
    out_r.cone_width = 0.0f;
    out_r.cone_spread = p.spread_angle;
    out_r.pdf = 1e6f;
    out_r.xy = uint32_t((x << 16) | y);
    out_r.depth = pack_ray_type(RAY_TYPE_CAMERA) | pack_ray_depth(0, 0, 0, 0);

    out_i = make_hit();
    out_i.t = (p.clip_end / dot(_d, p.fwd)) - clip_start;
}

} // namespace rt
