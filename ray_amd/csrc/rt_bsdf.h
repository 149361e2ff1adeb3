// rt_bsdf.h -- BSDF evaluation / importance sampling and the per-node Evaluate_/Sample_ pairs.
// Restates reference internal/ShadeRef.cpp:12-1028 (GLSL twin: shaders/shade.comp.glsl).  Function names keep
// the reference's so the two can be read side by side.  Return convention of the reference is kept too:
// BSDF functions return rgb in .xyz and the pdf in .w.
#pragma once

#include "rt_lights.h"
#include "rt_rng.h"
#include "rt_types.h"

namespace rt {

struct Surface { // CoreRef.h:108-113 surface_t
    f3 P, T, B, N, plane_N;
    f2 uvs;
};

struct LobeWeights { // ShadeRef.h lobe_weights_t
    float diffuse, specular, clearcoat, refraction;
};
struct DiffParams {
    f3 base_color;
    f3 sheen_color;
    float roughness;
};
struct SpecParams {
    f3 tmp_col;
    float roughness, ior, F0, anisotropy;
};
struct CoatParams {
    float roughness, ior, F0;
};
struct TransParams {
    float roughness, int_ior, eta, fresnel;
    bool backfacing;
};

// ShadeRef.cpp:12-20
RT_HD f2 calc_alpha(const float roughness, const float anisotropy, const float regularize_alpha) {
    const float roughness2 = sqr(roughness);
    const float aspect = sqrtf(1.0f - 0.9f * anisotropy);

    f2 alpha = {roughness2 / aspect, roughness2 * aspect};
    // where(alpha < reg, alpha) = clamp(2 * alpha, 0.25f * reg, reg); generic clamp = min(max(v, lo), hi)
    if (alpha.x < regularize_alpha) {
        const float v = 2 * alpha.x, lo = 0.25f * regularize_alpha;
        const float m = (v < lo) ? lo : v;                      // std::max(v, lo)
        alpha.x = (regularize_alpha < m) ? regularize_alpha : m; // std::min(m, hi)
    }
    if (alpha.y < regularize_alpha) {
        const float v = 2 * alpha.y, lo = 0.25f * regularize_alpha;
        const float m = (v < lo) ? lo : v;
        alpha.y = (regularize_alpha < m) ? regularize_alpha : m;
    }
    return alpha;
}

RT_HD float pow5(const float v) { return (v * v) * (v * v) * v; }
RT_HD float schlick_weight(const float u) {
    const float m = saturatef(1.0f - u);
    return pow5(m);
}
RT_HD f3 reflect(const f3 I, const f3 N, const float dot_N_I) { return I - 2 * dot_N_I * N; }

// ShadeRef.cpp:32-52
RT_HD LobeWeights get_lobe_weights(const float base_color_lum, const float spec_color_lum, const float specular,
                                   const float metallic, const float transmission, const float clearcoat) {
    LobeWeights weights;
    // taken from Cycles
    weights.diffuse = base_color_lum * (1.0f - metallic) * (1.0f - transmission);
    const float final_transmission = transmission * (1.0f - metallic);
    weights.specular = (specular != 0.0f || metallic != 0.0f) ? spec_color_lum * (1.0f - final_transmission) : 0.0f;
    weights.clearcoat = 0.25f * clearcoat * (1.0f - metallic);
    weights.refraction = final_transmission * base_color_lum;

    const float total_weight = weights.diffuse + weights.specular + weights.clearcoat + weights.refraction;
    if (total_weight != 0.0f) {
        weights.diffuse /= total_weight;
        weights.specular /= total_weight;
        weights.clearcoat /= total_weight;
        weights.refraction /= total_weight;
    }
    return weights;
}

// ShadeRef.cpp:54-71
RT_HD float fresnel_dielectric_cos(float cosi, float eta) {
    // compute fresnel reflectance without explicitly computing the refracted direction
    float c = fabsf(cosi);
    float g = eta * eta - 1 + c * c;
    float result;
    if (g > 0) {
        g = sqrtf(g);
        float A = (g - c) / (g + c);
        float B = (c * (g + c) - 1) / (c * (g - c) + 1);
        result = 0.5f * A * A * (1 + B * B);
    } else {
        result = 1.0f; // TIR (no refracted component)
    }
    return result;
}

// https://arxiv.org/pdf/2306.05044.pdf  ShadeRef.cpp:126-138
RT_HD f3 SampleVNDF_Hemisphere_SphCap(const f3 Vh, const f2 alpha, const f2 rnd) {
    // sample a spherical cap in (-Vh.z, 1]
    const float phi = 2.0f * PI * rnd.x;
    const float z = fmaf(1.0f - rnd.y, 1.0f + Vh.z, -Vh.z);
    const float sin_theta = sqrtf(saturatef(1.0f - z * z));
    const f2 sincos_phi = portable_sincos(phi);
    const float x = sin_theta * sincos_phi.y;
    const float y = sin_theta * sincos_phi.x;
    const f3 c = {x, y, z};
    // normalization will be done later
    return c + Vh;
}

// Bounded VNDF, ShadeRef.cpp:141-157
RT_HD f3 SampleVNDF_Hemisphere_SphCap_Bounded(const f3 Ve, const f3 Vh, const f2 alpha, const f2 rnd) {
    // sample a spherical cap in (-Vh.z, 1]
    const float phi = 2.0f * PI * rnd.x;
    const float a = saturatef(fminf(alpha.x, alpha.y));
    const float s = 1.0f + length(mk2(Ve.x, Ve.y));
    const float a2 = a * a, s2 = s * s;
    const float k = (1.0f - a2) * s2 / (s2 + a2 * Ve.z * Ve.z);
    const float b = (Ve.z > 0.0f) ? k * Vh.z : Vh.z;
    const float z = fmaf(1.0f - rnd.y, 1.0f + b, -b);
    const float sin_theta = sqrtf(saturatef(1.0f - z * z));
    const f2 sincos_phi = portable_sincos(phi);
    const float x = sin_theta * sincos_phi.y;
    const float y = sin_theta * sincos_phi.x;
    const f3 c = {x, y, z};
    // normalization will be done later
    return c + Vh;
}

// ShadeRef.cpp:163-184
RT_HD f3 SampleGGX_VNDF(const f3 Ve, f2 alpha, f2 rnd) {
    // transforming the view direction to the hemisphere configuration
    const f3 Vh = normalize(mk3(alpha.x * Ve.x, alpha.y * Ve.y, Ve.z));
    // sample the hemisphere
    const f3 Nh = SampleVNDF_Hemisphere_SphCap(Vh, alpha, rnd);
    // transforming the normal back to the ellipsoid configuration
    const f3 Ne = normalize(mk3(alpha.x * Nh.x, alpha.y * Nh.y, fmaxf(0.0f, Nh.z)));
    return Ne;
}
RT_HD f3 SampleGGX_VNDF_Bounded(const f3 Ve, f2 alpha, f2 rnd) {
    const f3 Vh = normalize(mk3(alpha.x * Ve.x, alpha.y * Ve.y, Ve.z));
    const f3 Nh = SampleVNDF_Hemisphere_SphCap_Bounded(Ve, Vh, alpha, rnd);
    const f3 Ne = normalize(mk3(alpha.x * Nh.x, alpha.y * Nh.y, fmaxf(0.0f, Nh.z)));
    return Ne;
}

// ShadeRef.cpp:186-198
RT_HD float GGX_VNDF_Reflection_Bounded_PDF(const float D, const f3 view_dir_ts, const f2 alpha) {
    const f2 ai = alpha * mk2(view_dir_ts.x, view_dir_ts.y);
    const float len2 = dot(ai, ai);
    const float t = sqrtf(len2 + view_dir_ts.z * view_dir_ts.z);
    if (view_dir_ts.z >= 0.0f) {
        const float a = saturatef(fminf(alpha.x, alpha.y));
        const float s = 1.0f + length(mk2(view_dir_ts.x, view_dir_ts.y));
        const float a2 = a * a, s2 = s * s;
        const float k = (1.0f - a2) * s2 / (s2 + a2 * view_dir_ts.z * view_dir_ts.z);
        return D / (2.0f * (k * view_dir_ts.z + t));
    }
    return D * (t - view_dir_ts.z) / (2.0f * len2);
}

// Smith shadowing function, ShadeRef.cpp:201-208
RT_HD float G1(const f3 Ve, f2 alpha) {
    alpha = alpha * alpha;
    const float delta =
        (-1.0f + sqrtf(1.0f + safe_div_pos(alpha.x * sqr(Ve.x) + alpha.y * sqr(Ve.y), sqr(Ve.z)))) / 2.0f;
    return 1.0f / (1.0f + delta);
}

// ShadeRef.cpp:216-223
RT_HD float D_GTR1(float NDotH, float a) {
    if (a >= 1.0f) {
        return 1.0f / PI;
    }
    const float a2 = sqr(a);
    const float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
    return (a2 - 1.0f) / (PI * logf(a2) * t);
}

// ShadeRef.cpp:231-240
RT_HD float D_GGX(const f3 H, const f2 alpha) {
    if (H.z == 0.0f) {
        return 0.0f;
    }
    const float sx = -H.x / (H.z * alpha.x);
    const float sy = -H.y / (H.z * alpha.y);
    const float s1 = 1.0f + sx * sx + sy * sy;
    const float cos_theta_h4 = sqr(sqr(H.z));
    return 1.0f / (sqr(s1) * PI * alpha.x * alpha.y * cos_theta_h4);
}

RT_HD float safe_sqrtf(float f) { return sqrtf(fmaxf(f, 0.0f)); }

// Taken from Cycles, ShadeRef.cpp:245-335
RT_HD f3 ensure_valid_reflection(const f3 Ng, const f3 I, const f3 N) {
    const f3 R = 2 * dot(N, I) * N - I;

    // Reflection rays may always be at least as shallow as the incoming ray.
    const float threshold = fminf(0.9f * dot(Ng, I), 0.01f);
    if (dot(Ng, R) >= threshold) {
        return N;
    }

    // Form coordinate system with Ng as the Z axis and N inside the X-Z-plane.
    const float NdotNg = dot(N, Ng);
    const f3 X = normalize(N - NdotNg * Ng);

    const float Ix = dot(I, X), Iz = dot(I, Ng);
    const float Ix2 = (Ix * Ix), Iz2 = (Iz * Iz);
    const float a = Ix2 + Iz2;

    const float b = safe_sqrtf(Ix2 * (a - (threshold * threshold)));
    const float c = Iz * threshold + a;

    const float fac = 0.5f / a;
    const float N1_z2 = fac * (b + c), N2_z2 = fac * (-b + c);
    bool valid1 = (N1_z2 > 1e-5f) && (N1_z2 <= (1.0f + 1e-5f));
    bool valid2 = (N2_z2 > 1e-5f) && (N2_z2 <= (1.0f + 1e-5f));

    f2 N_new;
    if (valid1 && valid2) {
        // If both are possible, do the expensive reflection-based check.
        const f2 N1 = {safe_sqrtf(1.0f - N1_z2), safe_sqrtf(N1_z2)};
        const f2 N2 = {safe_sqrtf(1.0f - N2_z2), safe_sqrtf(N2_z2)};

        const float R1 = 2 * (N1.x * Ix + N1.y * Iz) * N1.y - Iz;
        const float R2 = 2 * (N2.x * Ix + N2.y * Iz) * N2.y - Iz;

        valid1 = (R1 >= 1e-5f);
        valid2 = (R2 >= 1e-5f);
        if (valid1 && valid2) {
            N_new = (R1 < R2) ? N1 : N2;
        } else {
            N_new = (R1 > R2) ? N1 : N2;
        }
    } else if (valid1 || valid2) {
        // Only one solution passes the N'.z criterium, so pick that one.
        const float Nz2 = valid1 ? N1_z2 : N2_z2;
        N_new = {safe_sqrtf(1.0f - Nz2), safe_sqrtf(Nz2)};
    } else {
        return Ng;
    }
    return N_new.x * X + N_new.y * Ng;
}

// ShadeRef.cpp:337-358
RT_HD f3 rotate_around_axis(const f3 p, const f3 axis, const float angle) {
    const f2 sincos_theta = portable_sincos(angle);
    const float costheta = sincos_theta.y;
    const float sintheta = sincos_theta.x;
    f3 r;
    r.x = ((costheta + (1.0f - costheta) * axis.x * axis.x) * p.x) +
          (((1.0f - costheta) * axis.x * axis.y - axis.z * sintheta) * p.y) +
          (((1.0f - costheta) * axis.x * axis.z + axis.y * sintheta) * p.z);
    r.y = (((1.0f - costheta) * axis.x * axis.y + axis.z * sintheta) * p.x) +
          ((costheta + (1.0f - costheta) * axis.y * axis.y) * p.y) +
          (((1.0f - costheta) * axis.y * axis.z - axis.x * sintheta) * p.z);
    r.z = (((1.0f - costheta) * axis.x * axis.z - axis.y * sintheta) * p.x) +
          (((1.0f - costheta) * axis.y * axis.z + axis.x * sintheta) * p.y) +
          ((costheta + (1.0f - costheta) * axis.z * axis.z) * p.z);
    return r;
}

// ior stack, ShadeRef.cpp:360-391
RT_HD void push_ior_stack(float stack[4], const float val) {
    for (int i = 0; i < 3; ++i) {
        if (stack[i] < 0.0f) {
            stack[i] = val;
            return;
        }
    }
    // replace the last value regardless of sign
    stack[3] = val;
}
RT_HD float pop_ior_stack(float stack[4], const float default_value = 1.0f) {
    for (int i = 3; i >= 0; --i) {
        if (stack[i] > 0.0f) {
            const float ret = stack[i];
            stack[i] = -1.0f;
            return ret;
        }
    }
    return default_value;
}
RT_HD float peek_ior_stack(const float stack[4], bool skip_first, const float default_value = 1.0f) {
    for (int i = 3; i >= 0; --i) {
        if (stack[i] > 0.0f) {
            const bool skip = skip_first;
            skip_first = false;
            if (!skip) {
                return stack[i];
            }
        }
    }
    return default_value;
}

// ShadeRef.cpp:385-401
RT_HD float BRDF_PrincipledDiffuse(const f3 V, const f3 N, const f3 L, const f3 H, const float roughness) {
    const float N_dot_L = dot(N, L);
    const float N_dot_V = dot(N, V);
    if (N_dot_L <= 0.0f /*|| N_dot_V <= 0.0f*/) {
        return 0.0f;
    }
    const float FL = schlick_weight(N_dot_L);
    const float FV = schlick_weight(N_dot_V);

    const float L_dot_H = dot(L, H);
    const float Fd90 = 0.5f + 2.0f * L_dot_H * L_dot_H * roughness;
    const float Fd = mixf(1.0f, Fd90, FL) * mixf(1.0f, Fd90, FV);
    return Fd;
}

// ShadeRef.cpp:403-427
RT_HD f4 Evaluate_OrenDiffuse_BSDF(const f3 V, const f3 N, const f3 L, const float roughness, const f3 base_color) {
    const float sigma = roughness;
    const float div = 1.0f / (PI + ((3.0f * PI - 4.0f) / 6.0f) * sigma);

    const float a = 1.0f * div;
    const float b = sigma * div;

    const float nl = fmaxf(dot(N, L), 0.0f);
    const float nv = fmaxf(dot(N, V), 0.0f);
    float t = dot(L, V) - nl * nv;

    if (t > 0.0f) {
        t /= fmaxf(nl, nv) + FLT_MIN;
    }
    const float is = nl * (a + b * t);

    const f3 diff_col = is * base_color;
    return mk4(diff_col, 0.5f / PI);
}

// ShadeRef.cpp:429-441 (note: `rand0 * rand1` under the sqrt is what the reference does)
RT_HD f4 Sample_OrenDiffuse_BSDF(const f3 T, const f3 B, const f3 N, const f3 I, const float roughness,
                                 const f3 base_color, const f2 rnd, f3 &out_V) {
    const float phi = 2 * PI * rnd.y;
    const f2 sincos_phi = portable_sincos(phi);
    const float cos_phi = sincos_phi.y, sin_phi = sincos_phi.x;

    const float dir = sqrtf(1.0f - rnd.x * rnd.y);
    const f3 V = {dir * cos_phi, dir * sin_phi, rnd.x}; // in tangent-space

    out_V = world_from_tangent(T, B, N, V);
    return Evaluate_OrenDiffuse_BSDF(-I, N, out_V, roughness, base_color);
}

// ShadeRef.cpp:443-468
RT_HD f4 Evaluate_PrincipledDiffuse_BSDF(const f3 V, const f3 N, const f3 L, const float roughness, const f3 base_color,
                                         const f3 sheen_color, const bool uniform_sampling) {
    float weight, pdf;
    if (uniform_sampling) {
        weight = 2 * dot(N, L);
        pdf = 0.5f / PI;
    } else {
        weight = 1.0f;
        pdf = dot(N, L) / PI;
    }

    f3 H = normalize(L + V);
    if (dot(V, H) < 0.0f) {
        H = -H;
    }

    f3 diff_col = base_color * (weight * BRDF_PrincipledDiffuse(V, N, L, H, roughness));

    const float FH = PI * schlick_weight(dot(L, H));
    diff_col += FH * sheen_color;
    return mk4(diff_col, pdf);
}

// ShadeRef.cpp:470-491
RT_HD f4 Sample_PrincipledDiffuse_BSDF(const f3 T, const f3 B, const f3 N, const f3 I, const float roughness,
                                       const f3 base_color, const f3 sheen_color, const bool uniform_sampling,
                                       const f2 rnd, f3 &out_V) {
    const float phi = 2 * PI * rnd.y;
    const f2 sincos_phi = portable_sincos(phi);
    const float cos_phi = sincos_phi.y, sin_phi = sincos_phi.x;

    f3 V;
    if (uniform_sampling) {
        const float dir = sqrtf(1.0f - rnd.x * rnd.x);
        V = {dir * cos_phi, dir * sin_phi, rnd.x}; // in tangent-space
    } else {
        const float dir = sqrtf(rnd.x);
        const float k = sqrtf(1.0f - rnd.x);
        V = {dir * cos_phi, dir * sin_phi, k}; // in tangent-space
    }
    out_V = world_from_tangent(T, B, N, V);
    return Evaluate_PrincipledDiffuse_BSDF(-I, N, out_V, roughness, base_color, sheen_color, uniform_sampling);
}

// ShadeRef.cpp:493-512
RT_HD f4 Evaluate_GGXSpecular_BSDF(const f3 view_dir_ts, const f3 sampled_normal_ts, const f3 reflected_dir_ts,
                                   const f2 alpha, const float spec_ior, const float spec_F0, const f3 spec_col,
                                   const f3 spec_col_90) {
    const float D = D_GGX(sampled_normal_ts, alpha);
    const float G = G1(view_dir_ts, alpha) * G1(reflected_dir_ts, alpha);

    const float FH = (fresnel_dielectric_cos(dot(view_dir_ts, sampled_normal_ts), spec_ior) - spec_F0) / (1.0f - spec_F0);
    f3 F = mix3(spec_col, spec_col_90, FH);

    const float denom = 4.0f * fabsf(view_dir_ts.z * reflected_dir_ts.z);
    F *= (denom != 0.0f) ? (D * G / denom) : 0.0f;
    F *= fmaxf(reflected_dir_ts.z, 0.0f);

    const float pdf = GGX_VNDF_Reflection_Bounded_PDF(D, view_dir_ts, alpha);
    return mk4(F, pdf);
}

// ShadeRef.cpp:514-536
RT_HD f4 Sample_GGXSpecular_BSDF(const f3 T, const f3 B, const f3 N, const f3 I, const f2 alpha, const float spec_ior,
                                 const float spec_F0, const f3 spec_col, const f3 spec_col_90, const f2 rnd, f3 &out_V) {
    if (alpha.x * alpha.y < 1e-7f) {
        const f3 V = reflect(I, N, dot(N, I));
        const float FH = (fresnel_dielectric_cos(dot(V, N), spec_ior) - spec_F0) / (1.0f - spec_F0);
        const f3 F = mix3(spec_col, spec_col_90, FH);
        out_V = V;
        return mk4(F.x * 1e6f, F.y * 1e6f, F.z * 1e6f, 1e6f);
    }

    const f3 view_dir_ts = normalize(tangent_from_world(T, B, N, -I));
    const f3 sampled_normal_ts = SampleGGX_VNDF_Bounded(view_dir_ts, alpha, rnd);

    const float dot_N_V = -dot(sampled_normal_ts, view_dir_ts);
    const f3 reflected_dir_ts = normalize(reflect(-view_dir_ts, sampled_normal_ts, dot_N_V));

    out_V = world_from_tangent(T, B, N, reflected_dir_ts);
    return Evaluate_GGXSpecular_BSDF(view_dir_ts, sampled_normal_ts, reflected_dir_ts, alpha, spec_ior, spec_F0, spec_col,
                                     spec_col_90);
}

// ShadeRef.cpp:538-568
RT_HD f4 Evaluate_GGXRefraction_BSDF(const f3 view_dir_ts, const f3 sampled_normal_ts, const f3 refr_dir_ts, const f2 alpha,
                                     float eta, const f3 refr_col) {
    if (refr_dir_ts.z >= 0.0f || view_dir_ts.z <= 0.0f || alpha.x * alpha.y < 1e-7f) {
        return mk4(0.0f, 0.0f, 0.0f, 0.0f);
    }

    const float D = D_GGX(sampled_normal_ts, alpha);

    const float G1o = G1(refr_dir_ts, alpha), G1i = G1(view_dir_ts, alpha);

    const float denom = dot(refr_dir_ts, sampled_normal_ts) + dot(view_dir_ts, sampled_normal_ts) * eta;
    const float jacobian = safe_div_pos(fmaxf(-dot(refr_dir_ts, sampled_normal_ts), 0.0f), denom * denom);

    const float F = D * G1i * G1o * fmaxf(dot(view_dir_ts, sampled_normal_ts), 0.0f) * jacobian / (view_dir_ts.z);

    const float pdf = D * G1o * fmaxf(dot(view_dir_ts, sampled_normal_ts), 0.0f) * jacobian / view_dir_ts.z;

    const f3 ret = F * refr_col;
    return mk4(ret, pdf);
}

// ShadeRef.cpp:570-606.  out_V.w carries `m` like the reference (unused by callers)
RT_HD f4 Sample_GGXRefraction_BSDF(const f3 T, const f3 B, const f3 N, const f3 I, const f2 alpha, const float eta,
                                   const f3 refr_col, const f2 rnd, f3 &out_V) {
    if (alpha.x * alpha.y < 1e-7f) {
        const float cosi = -dot(I, N);
        const float cost2 = 1.0f - eta * eta * (1.0f - cosi * cosi);
        if (cost2 < 0) {
            return mk4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        const float m = eta * cosi - sqrtf(cost2);
        const f3 V = normalize(eta * I + m * N);

        out_V = V;
        return mk4(refr_col.x * 1e6f, refr_col.y * 1e6f, refr_col.z * 1e6f, 1e6f);
    }

    const f3 view_dir_ts = normalize(tangent_from_world(T, B, N, -I));
    const f3 sampled_normal_ts = SampleGGX_VNDF(view_dir_ts, alpha, rnd);

    const float cosi = dot(view_dir_ts, sampled_normal_ts);
    const float cost2 = 1.0f - eta * eta * (1.0f - cosi * cosi);
    if (cost2 < 0) {
        return mk4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    const float m = eta * cosi - sqrtf(cost2);
    const f3 refr_dir_ts = normalize(-eta * view_dir_ts + m * sampled_normal_ts);

    const f4 F = Evaluate_GGXRefraction_BSDF(view_dir_ts, sampled_normal_ts, refr_dir_ts, alpha, eta, refr_col);

    out_V = world_from_tangent(T, B, N, refr_dir_ts);
    return F;
}

// ShadeRef.cpp:608-628
RT_HD f4 Evaluate_PrincipledClearcoat_BSDF(const f3 view_dir_ts, const f3 sampled_normal_ts, const f3 reflected_dir_ts,
                                           const float clearcoat_roughness2, const float clearcoat_ior,
                                           const float clearcoat_F0) {
    const float D = D_GTR1(sampled_normal_ts.z, clearcoat_roughness2);
    // Always assume roughness of 0.25 for clearcoat
    const f2 clearcoat_alpha = {0.25f * 0.25f, 0.25f * 0.25f};
    const float G = G1(view_dir_ts, clearcoat_alpha) * G1(reflected_dir_ts, clearcoat_alpha);

    const float FH = (fresnel_dielectric_cos(dot(reflected_dir_ts, sampled_normal_ts), clearcoat_ior) - clearcoat_F0) /
                     (1.0f - clearcoat_F0);
    float F = mixf(0.04f, 1.0f, FH);

    const float denom = 4.0f * fabsf(view_dir_ts.z) * fabsf(reflected_dir_ts.z);
    F *= (denom != 0.0f) ? D * G / denom : 0.0f;
    F *= fmaxf(reflected_dir_ts.z, 0.0f);

    const float pdf = GGX_VNDF_Reflection_Bounded_PDF(D, view_dir_ts, clearcoat_alpha);
    return mk4(F, F, F, pdf);
}

// ShadeRef.cpp:630-657
RT_HD f4 Sample_PrincipledClearcoat_BSDF(const f3 T, const f3 B, const f3 N, const f3 I, const float clearcoat_roughness2,
                                         const float clearcoat_ior, const float clearcoat_F0, const f2 rnd, f3 &out_V) {
    if (sqr(clearcoat_roughness2) < 1e-7f) {
        const f3 V = reflect(I, N, dot(N, I));

        const float FH = (fresnel_dielectric_cos(dot(V, N), clearcoat_ior) - clearcoat_F0) / (1.0f - clearcoat_F0);
        const float F = mixf(0.04f, 1.0f, FH);

        out_V = V;
        return mk4(F * 1e6f, F * 1e6f, F * 1e6f, 1e6f);
    }

    const f3 view_dir_ts = normalize(tangent_from_world(T, B, N, -I));
    // NOTE: GTR1 distribution is not used for sampling because Cycles does it this way (???!)
    const f3 sampled_normal_ts =
        SampleGGX_VNDF_Bounded(view_dir_ts, mk2(clearcoat_roughness2, clearcoat_roughness2), rnd);

    const float dot_N_V = -dot(sampled_normal_ts, view_dir_ts);
    const f3 reflected_dir_ts = normalize(reflect(-view_dir_ts, sampled_normal_ts, dot_N_V));

    out_V = world_from_tangent(T, B, N, reflected_dir_ts);

    return Evaluate_PrincipledClearcoat_BSDF(view_dir_ts, sampled_normal_ts, reflected_dir_ts, clearcoat_roughness2,
                                             clearcoat_ior, clearcoat_F0);
}

// ---- shading nodes ------------------------------------------------------------------------------------
// Evaluate_* return the light contribution to add right away (lights that cast no shadow) and otherwise fill
// sh_r.o / sh_r.c for the shadow ray and return 0.

// ShadeRef.cpp:659-683
RT_HD f3 Evaluate_DiffuseNode(const LightSample &ls, const Ray &ray, const Surface &surf, const f3 base_color,
                              const float roughness, const float mix_weight, const bool use_mis, ShadowRay &sh_r) {
    const f3 I = ray.d;

    const f4 diff_col = Evaluate_OrenDiffuse_BSDF(-I, surf.N, ls.L, roughness, base_color);
    const float bsdf_pdf = diff_col.w;

    float mis_weight = 1.0f;
    if (use_mis && ls.area > 0.0f) {
        mis_weight = power_heuristic(ls.pdf, bsdf_pdf);
    }
    const f3 lcol = ls.col * xyz(diff_col) * (mix_weight * mis_weight / ls.pdf);

    if (!ls.cast_shadow) {
        // apply light immediately
        return lcol;
    }
    // schedule shadow ray
    sh_r.o = offset_ray(surf.P, surf.plane_N);
    sh_r.c = lcol;
    return f3{0.0f, 0.0f, 0.0f};
}

// ShadeRef.cpp:685-702
RT_HD void Sample_DiffuseNode(const Ray &ray, const Surface &surf, const f3 base_color, const float roughness, const f2 rnd,
                              const float mix_weight, Ray &new_ray) {
    const f3 I = ray.d;

    f3 V;
    const f4 F = Sample_OrenDiffuse_BSDF(surf.T, surf.B, surf.N, I, roughness, base_color, rnd, V);

    new_ray.depth = pack_ray_type(RAY_TYPE_DIFFUSE);
    new_ray.depth |= mask_ray_depth(ray.depth) + pack_ray_depth(1, 0, 0, 0);

    new_ray.o = offset_ray(surf.P, surf.plane_N);
    new_ray.d = V;
    new_ray.c = {F.x * mix_weight / F.w, F.y * mix_weight / F.w, F.z * mix_weight / F.w};
    new_ray.pdf = F.w;
    new_ray.cone_spread += MAX_CONE_SPREAD_INCREMENT;
}

// ShadeRef.cpp:704-741
RT_HD f3 Evaluate_GlossyNode(const LightSample &ls, const Ray &ray, const Surface &surf, const f3 base_color,
                             const float roughness, const float regularize_alpha, const float spec_ior, const float spec_F0,
                             const float mix_weight, const bool use_mis, ShadowRay &sh_r) {
    const f3 I = ray.d;
    const f3 H = normalize(ls.L - I);

    const f3 view_dir_ts = tangent_from_world(surf.T, surf.B, surf.N, -I);
    const f3 light_dir_ts = tangent_from_world(surf.T, surf.B, surf.N, ls.L);
    const f3 sampled_normal_ts = tangent_from_world(surf.T, surf.B, surf.N, H);

    const f2 alpha = calc_alpha(roughness, 0.0f, regularize_alpha);
    if (alpha.x * alpha.y < 1e-7f) {
        return f3{0.0f, 0.0f, 0.0f};
    }

    const f4 spec_col = Evaluate_GGXSpecular_BSDF(view_dir_ts, sampled_normal_ts, light_dir_ts, alpha, spec_ior, spec_F0,
                                                  base_color, base_color);
    const float bsdf_pdf = spec_col.w;

    float mis_weight = 1.0f;
    if (use_mis && ls.area > 0.0f) {
        mis_weight = power_heuristic(ls.pdf, bsdf_pdf);
    }
    const f3 lcol = ls.col * xyz(spec_col) * (mix_weight * mis_weight / ls.pdf);

    if (!ls.cast_shadow) {
        return lcol;
    }
    sh_r.o = offset_ray(surf.P, surf.plane_N);
    sh_r.c = lcol;
    return f3{0.0f, 0.0f, 0.0f};
}

// ShadeRef.cpp:743-763
RT_HD void Sample_GlossyNode(const Ray &ray, const Surface &surf, const f3 base_color, const float roughness,
                             const float regularize_alpha, const float spec_ior, const float spec_F0, const f2 rnd,
                             const float mix_weight, Ray &new_ray) {
    const f3 I = ray.d;
    const f2 alpha = calc_alpha(roughness, 0.0f, regularize_alpha);

    f3 V;
    const f4 F = Sample_GGXSpecular_BSDF(surf.T, surf.B, surf.N, I, alpha, spec_ior, spec_F0, base_color, base_color, rnd, V);

    new_ray.depth = pack_ray_type(RAY_TYPE_SPECULAR);
    new_ray.depth |= mask_ray_depth(ray.depth) + pack_ray_depth(0, 1, 0, 0);

    new_ray.o = offset_ray(surf.P, surf.plane_N);
    new_ray.d = V;

    const float k = safe_div_pos(mix_weight, F.w);
    new_ray.c = {F.x * k, F.y * k, F.z * k};
    new_ray.pdf = F.w;
    new_ray.cone_spread += MAX_CONE_SPREAD_INCREMENT * fminf(alpha.x, alpha.y);
}

// ShadeRef.cpp:765-796
RT_HD f3 Evaluate_RefractiveNode(const LightSample &ls, const Ray &ray, const Surface &surf, const f3 base_color,
                                 const float roughness, const float regularize_alpha, const float eta,
                                 const float mix_weight, const bool use_mis, ShadowRay &sh_r) {
    const f3 I = ray.d;

    const f3 H = normalize(ls.L - I * eta);
    const f3 view_dir_ts = tangent_from_world(surf.T, surf.B, surf.N, -I);
    const f3 light_dir_ts = tangent_from_world(surf.T, surf.B, surf.N, ls.L);
    const f3 sampled_normal_ts = tangent_from_world(surf.T, surf.B, surf.N, H);

    const f4 refr_col = Evaluate_GGXRefraction_BSDF(view_dir_ts, sampled_normal_ts, light_dir_ts,
                                                    calc_alpha(roughness, 0.0f, regularize_alpha), eta, base_color);
    const float bsdf_pdf = refr_col.w;

    float mis_weight = 1.0f;
    if (use_mis && ls.area > 0.0f) {
        mis_weight = power_heuristic(ls.pdf, bsdf_pdf);
    }
    const f3 lcol = ls.col * xyz(refr_col) * (mix_weight * mis_weight / ls.pdf);

    if (!ls.cast_shadow) {
        return lcol;
    }
    sh_r.o = offset_ray(surf.P, -surf.plane_N);
    sh_r.c = lcol;
    return f3{0.0f, 0.0f, 0.0f};
}

// ShadeRef.cpp:780-808
RT_HD void Sample_RefractiveNode(const Ray &ray, const Surface &surf, const f3 base_color, const float roughness,
                                 const float regularize_alpha, const bool is_backfacing, const float int_ior,
                                 const float ext_ior, const f2 rnd, const float mix_weight, Ray &new_ray) {
    const f3 I = ray.d;
    const f2 alpha = calc_alpha(roughness, 0.0f, regularize_alpha);
    const float eta = is_backfacing ? (int_ior / ext_ior) : (ext_ior / int_ior);

    f3 V;
    const f4 F = Sample_GGXRefraction_BSDF(surf.T, surf.B, surf.N, I, alpha, eta, base_color, rnd, V);

    new_ray.depth = pack_ray_type(RAY_TYPE_REFR);
    new_ray.depth |= mask_ray_depth(ray.depth) + pack_ray_depth(0, 0, 1, 0);

    const float k = safe_div_pos(mix_weight, F.w);
    new_ray.c = {F.x * k, F.y * k, F.z * k};
    new_ray.pdf = F.w;

    if (!is_backfacing) {
        // Entering the surface, push new value
        push_ior_stack(new_ray.ior, int_ior);
    } else {
        // Exiting the surface, pop the last ior value
        pop_ior_stack(new_ray.ior);
    }

    new_ray.o = offset_ray(surf.P, -surf.plane_N);
    new_ray.d = V;
    new_ray.cone_spread += MAX_CONE_SPREAD_INCREMENT * fminf(alpha.x, alpha.y);
}

// ShadeRef.cpp:810-901
RT_HD f3 Evaluate_PrincipledNode(const LightSample &ls, const Ray &ray, const Surface &surf, const LobeWeights &lobe_weights,
                                 const DiffParams &diff, const SpecParams &spec, const CoatParams &coat,
                                 const TransParams &trans, const float metallic, const float transmission,
                                 const float N_dot_L, const float mix_weight, const bool use_mis,
                                 const float regularize_alpha, ShadowRay &sh_r) {
    const f3 I = ray.d;

    f3 lcol = {0.0f, 0.0f, 0.0f};
    float bsdf_pdf = 0.0f;

    if (lobe_weights.diffuse > 0.0f && N_dot_L > 0.0f && (ls.ray_flags & RAY_TYPE_DIFFUSE_BIT) != 0) {
        f4 diff_col = Evaluate_PrincipledDiffuse_BSDF(-I, surf.N, ls.L, diff.roughness, diff.base_color, diff.sheen_color, false);
        bsdf_pdf += lobe_weights.diffuse * diff_col.w;
        const f3 dc = xyz(diff_col) * ((1.0f - metallic) * (1.0f - transmission));

        lcol += ls.col * N_dot_L * dc / (PI * ls.pdf);
    }

    f3 H;
    if (N_dot_L > 0.0f) {
        H = normalize(ls.L - I);
    } else {
        H = normalize(ls.L - I * trans.eta);
    }

    const f3 view_dir_ts = tangent_from_world(surf.T, surf.B, surf.N, -I);
    const f3 light_dir_ts = tangent_from_world(surf.T, surf.B, surf.N, ls.L);
    const f3 sampled_normal_ts = tangent_from_world(surf.T, surf.B, surf.N, H);

    const f2 spec_alpha = calc_alpha(spec.roughness, spec.anisotropy, regularize_alpha);
    if (lobe_weights.specular > 0.0f && spec_alpha.x * spec_alpha.y >= 1e-7f && N_dot_L > 0.0f &&
        (ls.ray_flags & RAY_TYPE_SPECULAR_BIT) != 0) {
        const f4 spec_col = Evaluate_GGXSpecular_BSDF(view_dir_ts, sampled_normal_ts, light_dir_ts, spec_alpha, spec.ior,
                                                      spec.F0, spec.tmp_col, splat3(1.0f));
        bsdf_pdf += lobe_weights.specular * spec_col.w;

        lcol += ls.col * xyz(spec_col) / ls.pdf;
    }

    const f2 coat_alpha = calc_alpha(coat.roughness, 0.0f, regularize_alpha);
    if (lobe_weights.clearcoat > 0.0f && coat_alpha.x * coat_alpha.y >= 1e-7f && N_dot_L > 0.0f &&
        (ls.ray_flags & RAY_TYPE_SPECULAR_BIT) != 0) {
        const f4 clearcoat_col =
            Evaluate_PrincipledClearcoat_BSDF(view_dir_ts, sampled_normal_ts, light_dir_ts, coat_alpha.x, coat.ior, coat.F0);
        bsdf_pdf += lobe_weights.clearcoat * clearcoat_col.w;

        lcol += 0.25f * ls.col * xyz(clearcoat_col) / ls.pdf;
    }

    if (lobe_weights.refraction > 0.0f) {
        const f2 refr_spec_alpha = calc_alpha(spec.roughness, 0.0f, regularize_alpha);
        if (trans.fresnel != 0.0f && refr_spec_alpha.x * refr_spec_alpha.y >= 1e-7f && N_dot_L > 0.0f &&
            (ls.ray_flags & RAY_TYPE_SPECULAR_BIT) != 0) {
            const f4 spec_col = Evaluate_GGXSpecular_BSDF(view_dir_ts, sampled_normal_ts, light_dir_ts, refr_spec_alpha,
                                                          1.0f /* ior */, 0.0f /* F0 */, splat3(1.0f), splat3(1.0f));
            bsdf_pdf += lobe_weights.refraction * trans.fresnel * spec_col.w;

            lcol += ls.col * xyz(spec_col) * (trans.fresnel / ls.pdf);
        }

        const f2 refr_trans_alpha = calc_alpha(trans.roughness, 0.0f, regularize_alpha);
        if (trans.fresnel != 1.0f && refr_trans_alpha.x * refr_trans_alpha.y >= 1e-7f && N_dot_L < 0.0f &&
            (ls.ray_flags & RAY_TYPE_REFR_BIT) != 0) {
            const f4 refr_col = Evaluate_GGXRefraction_BSDF(view_dir_ts, sampled_normal_ts, light_dir_ts, refr_trans_alpha,
                                                            trans.eta, diff.base_color);
            bsdf_pdf += lobe_weights.refraction * (1.0f - trans.fresnel) * refr_col.w;

            lcol += ls.col * xyz(refr_col) * ((1.0f - trans.fresnel) / ls.pdf);
        }
    }

    float mis_weight = 1.0f;
    if (use_mis && ls.area > 0.0f) {
        mis_weight = power_heuristic(ls.pdf, bsdf_pdf);
    }
    lcol *= mix_weight * mis_weight;

    if (!ls.cast_shadow) {
        return lcol;
    }
    // schedule shadow ray
    sh_r.o = offset_ray(surf.P, N_dot_L < 0.0f ? -surf.plane_N : surf.plane_N);
    sh_r.c = lcol;
    return f3{0.0f, 0.0f, 0.0f};
}

struct PassLimits {
    int max_diff_depth, max_spec_depth, max_refr_depth, max_transp_depth, max_total_depth;
    int min_total_depth, min_transp_depth;
    float regularize_alpha;
};

// ShadeRef.cpp:903-1026
RT_HD void Sample_PrincipledNode(const PassLimits &ps, const Ray &ray, const Surface &surf, const LobeWeights &lobe_weights,
                                 const DiffParams &diff, const SpecParams &spec, const CoatParams &coat,
                                 const TransParams &trans, const float metallic, const float transmission, const f2 rnd,
                                 float mix_rand, const float mix_weight, const float regularize_alpha, Ray &new_ray) {
    const f3 I = ray.d;

    const int diff_depth = get_diff_depth(ray.depth), spec_depth = get_spec_depth(ray.depth),
              refr_depth = get_refr_depth(ray.depth);
    // NOTE: transparency depth is not accounted here
    const int total_depth = diff_depth + spec_depth + refr_depth;

    if (mix_rand < lobe_weights.diffuse) {
        // Diffuse lobe
        if (diff_depth < ps.max_diff_depth && total_depth < ps.max_total_depth) {
            f3 V;
            f4 F = Sample_PrincipledDiffuse_BSDF(surf.T, surf.B, surf.N, I, diff.roughness, diff.base_color, diff.sheen_color,
                                                 false, rnd, V);
            const float pdf = F.w; // * lobe_weights.diffuse;

            F *= (1.0f - metallic) * (1.0f - transmission);

            new_ray.depth = pack_ray_type(RAY_TYPE_DIFFUSE);
            new_ray.depth |= mask_ray_depth(ray.depth) + pack_ray_depth(1, 0, 0, 0);

            new_ray.o = offset_ray(surf.P, surf.plane_N);
            new_ray.d = V;

            const float k = safe_div_pos(mix_weight, lobe_weights.diffuse);
            new_ray.c = {F.x * k, F.y * k, F.z * k};
            new_ray.pdf = pdf;
            new_ray.cone_spread += MAX_CONE_SPREAD_INCREMENT;
        }
    } else if (mix_rand < lobe_weights.diffuse + lobe_weights.specular) {
        // Main specular lobe
        if (spec_depth < ps.max_spec_depth && total_depth < ps.max_total_depth) {
            const f2 alpha = calc_alpha(spec.roughness, spec.anisotropy, regularize_alpha);
            f3 V;
            const f4 F = Sample_GGXSpecular_BSDF(surf.T, surf.B, surf.N, I, alpha, spec.ior, spec.F0, spec.tmp_col,
                                                 splat3(1.0f), rnd, V);
            const float pdf = F.w * lobe_weights.specular;

            new_ray.depth = pack_ray_type(RAY_TYPE_SPECULAR);
            new_ray.depth |= mask_ray_depth(ray.depth) + pack_ray_depth(0, 1, 0, 0);

            const float k = safe_div_pos(mix_weight, pdf);
            new_ray.c = {F.x * k, F.y * k, F.z * k};
            new_ray.pdf = pdf;

            new_ray.o = offset_ray(surf.P, surf.plane_N);
            new_ray.d = V;
            new_ray.cone_spread += MAX_CONE_SPREAD_INCREMENT * fminf(alpha.x, alpha.y);
        }
    } else if (mix_rand < lobe_weights.diffuse + lobe_weights.specular + lobe_weights.clearcoat) {
        // Clearcoat lobe (secondary specular)
        if (spec_depth < ps.max_spec_depth && total_depth < ps.max_total_depth) {
            const float alpha = calc_alpha(coat.roughness, 0.0f, regularize_alpha).x;
            f3 V;
            const f4 F = Sample_PrincipledClearcoat_BSDF(surf.T, surf.B, surf.N, I, alpha, coat.ior, coat.F0, rnd, V);
            const float pdf = F.w * lobe_weights.clearcoat;

            new_ray.depth = pack_ray_type(RAY_TYPE_SPECULAR);
            new_ray.depth |= mask_ray_depth(ray.depth) + pack_ray_depth(0, 1, 0, 0);

            const float k = safe_div_pos(mix_weight, pdf);
            new_ray.c = {0.25f * F.x * k, 0.25f * F.y * k, 0.25f * F.z * k};
            new_ray.pdf = pdf;

            new_ray.o = offset_ray(surf.P, surf.plane_N);
            new_ray.d = V;
            new_ray.cone_spread += MAX_CONE_SPREAD_INCREMENT * alpha;
        }
    } else {
        // Refraction/reflection lobes
        mix_rand -= lobe_weights.diffuse + lobe_weights.specular + lobe_weights.clearcoat;
        mix_rand = safe_div_pos(mix_rand, lobe_weights.refraction);
        if (((mix_rand >= trans.fresnel && refr_depth < ps.max_refr_depth) ||
             (mix_rand < trans.fresnel && spec_depth < ps.max_spec_depth)) &&
            total_depth < ps.max_total_depth) {
            f4 F;
            f3 V;
            if (mix_rand < trans.fresnel) {
                const f2 alpha = calc_alpha(spec.roughness, 0.0f, regularize_alpha);
                F = Sample_GGXSpecular_BSDF(surf.T, surf.B, surf.N, I, alpha, 1.0f /* ior */, 0.0f /* F0 */, splat3(1.0f),
                                            splat3(1.0f), rnd, V);

                new_ray.depth = pack_ray_type(RAY_TYPE_SPECULAR);
                new_ray.depth |= mask_ray_depth(ray.depth) + pack_ray_depth(0, 1, 0, 0);
                new_ray.o = offset_ray(surf.P, surf.plane_N);
                new_ray.cone_spread += MAX_CONE_SPREAD_INCREMENT * fminf(alpha.x, alpha.y);
            } else {
                const f2 alpha = calc_alpha(trans.roughness, 0.0f, regularize_alpha);
                F = Sample_GGXRefraction_BSDF(surf.T, surf.B, surf.N, I, alpha, trans.eta, diff.base_color, rnd, V);

                new_ray.depth = pack_ray_type(RAY_TYPE_REFR);
                new_ray.depth |= mask_ray_depth(ray.depth) + pack_ray_depth(0, 0, 1, 0);
                new_ray.o = offset_ray(surf.P, -surf.plane_N);
                new_ray.cone_spread += MAX_CONE_SPREAD_INCREMENT * fminf(alpha.x, alpha.y);

                if (!trans.backfacing) {
                    // Entering the surface, push new value
                    push_ior_stack(new_ray.ior, trans.int_ior);
                } else {
                    // Exiting the surface, pop the last ior value
                    pop_ior_stack(new_ray.ior);
                }
            }

            const float pdf = F.w * lobe_weights.refraction;

            const float k = safe_div_pos(mix_weight, pdf);
            new_ray.c = {F.x * k, F.y * k, F.z * k};
            new_ray.pdf = pdf;
            new_ray.d = V;
        }
    }
}

} // namespace rt
