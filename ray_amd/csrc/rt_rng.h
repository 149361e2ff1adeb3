// rt_rng.h -- integer hashing / Owen-scrambled PMJ02 lookup and the reference's polynomial trig.
// The integer part must match the reference bit for bit.
#pragma once

#include "rt_base.h"

namespace rt {

// CoreRef.h:133-141 murmur3 finalizer
RT_HD uint32_t hash(uint32_t x) {
    x ^= x >> 16;
    x *= 0x85ebca6bu;
    x ^= x >> 13;
    x *= 0xc2b2ae35u;
    x ^= x >> 16;
    return x;
}
// CoreRef.h:143
RT_HD uint32_t hash_combine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); }

// CoreRef.cpp:1068-1074
RT_HD uint32_t reverse_bits(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(x); // v_bfrev_b32: same permutation as the shift/mask ladder
#else
    x = (((x & 0xaaaaaaaa) >> 1) | ((x & 0x55555555) << 1));
    x = (((x & 0xcccccccc) >> 2) | ((x & 0x33333333) << 2));
    x = (((x & 0xf0f0f0f0) >> 4) | ((x & 0x0f0f0f0f) << 4));
    x = (((x & 0xff00ff00) >> 8) | ((x & 0x00ff00ff) << 8));
    return ((x >> 16) | (x << 16));
#endif
}
// CoreRef.cpp:1076-1083
RT_HD uint32_t laine_karras_permutation(uint32_t x, uint32_t seed) {
    x += seed;
    x ^= x * 0x6c50b47cu;
    x ^= x * 0xb82f1e52u;
    x ^= x * 0xc7afe638u;
    x ^= x * 0x8d22f6e6u;
    return x;
}
// CoreRef.cpp:1085-1090
RT_HD uint32_t nested_uniform_scramble_base2(uint32_t x, uint32_t seed) {
    x = reverse_bits(x);
    x = laine_karras_permutation(x, seed);
    x = reverse_bits(x);
    return x;
}
// CoreRef.cpp:1098-1101
RT_HD float scramble_unorm(uint32_t seed, uint32_t val) {
    val = nested_uniform_scramble_base2(val, seed);
    return float(val >> 8) / 16777216.0f;
}
// CoreRef.cpp:1418-1427
RT_HD f2 get_scrambled_2d_rand(uint32_t dim, uint32_t seed, int sample, const uint32_t *rand_seq) {
    const uint32_t shuffled_dim = nested_uniform_scramble_base2(dim, seed) & (RAND_DIMS_COUNT - 1);
    const uint32_t shuffled_i =
        nested_uniform_scramble_base2(uint32_t(sample), hash_combine(seed, dim)) & (RAND_SAMPLES_COUNT - 1);
    const uint32_t base = shuffled_dim * 2 * RAND_SAMPLES_COUNT + 2 * shuffled_i;
    return f2{scramble_unorm(hash_combine(seed, 2 * dim + 0), rand_seq[base + 0]),
              scramble_unorm(hash_combine(seed, 2 * dim + 1), rand_seq[base + 1])};
}

// ---- polynomial trig, CoreRef.cpp:1131-1231 ---------------------------------------------------------
// The reference evaluates three range candidates in SSE lanes and blends them with a 0/1 selector through
// dot(); here the same candidate polynomial is evaluated only for the selected range, and the blend
// collapses to res*1 (+0 terms), which is exact.  selector = {a<0.25, -1+s0+s2, a>=0.75, 0}:
//   a < 0.25        -> s = { 1, 0, 0}  -> +poly((a-0)^2)
//   0.25<=a<0.75    -> s = { 0,-1, 0}  -> -poly((a-0.5)^2)
//   a >= 0.75       -> s = { 0, 0, 1}  -> +poly((a-1)^2)
// dot() = (r0*s0 + r1*s1) + (r3*0 + r2*s2): the unselected products are +-0 and do not perturb the sum
// (poly is finite).
RT_HD float trig_poly_(float arg) {
    arg *= arg;
    float res = -25.0407296503853054f * arg + 60.1524123580209817f;
    res = res * arg - 85.4539888046442542f;
    res = res * arg + 64.9393549651994562f;
    res = res * arg - 19.7392086060579359f;
    res = res * arg + 0.9999999998415476f;
    return res;
}
RT_HD float trig_select_(float a) {
    if (a < 0.25f) {
        return trig_poly_(a - 0.0f);
    } else if (a >= 0.75f) {
        return trig_poly_(a - 1.0f);
    }
    return -trig_poly_(a - 0.5f);
}
RT_HD float portable_cos(float a) { return trig_select_(fractf(fabsf(a) * 0.15915494309189535f)); }
RT_HD float portable_sin(float a) {
    return trig_select_(fractf(fabsf(a - 1.5707963267948966f) * 0.15915494309189535f));
}
// returns {sin, cos} like the reference's fvec2
RT_HD f2 portable_sincos(float a) { return f2{portable_sin(a), portable_cos(a)}; }

// CoreRef.cpp:1233-1272 (from apple libm)
RT_HD float asin_tail(float x) {
    return (PI / 2) - ((x + 2.71745038f) * x + 14.0375338f) * (0.00440413551f * ((x - 8.31223679f) * x + 25.3978882f)) *
                          sqrtf(1 - x);
}
RT_HD float portable_asinf(float x) {
    if (fabsf(x) > 0.57f) {
        const float ret = asin_tail(fabsf(x));
        return (x < 0.0f) ? -ret : ret;
    } else {
        const float x2 = x * x;
        return x + (0.0517513789f * ((x2 + 1.83372748f) * x2 + 1.56678128f)) * x *
                       (x2 * ((x2 - 1.48268414f) * x2 + 2.05554748f));
    }
}
RT_HD float acos_positive_tail(float x) {
    return (((x + 2.71850395f) * x + 14.7303705f)) * (0.00393401226f * ((x - 8.60734272f) * x + 27.0927486f)) *
           sqrtf(1 - x);
}
RT_HD float acos_negative_tail(float x) {
    return PI - (((x - 2.71850395f) * x + 14.7303705f)) * (0.00393401226f * ((x + 8.60734272f) * x + 27.0927486f)) *
                    sqrtf(1 + x);
}
RT_HD float portable_acosf(float x) {
    if (x < -0.62f) {
        return acos_negative_tail(x);
    } else if (x <= 0.62f) {
        const float x2 = x * x;
        return (PI / 2) - x -
               (0.0700945929f * x * ((x2 + 1.57144082f) * x2 + 1.25210774f)) *
                   (x2 * ((x2 - 1.53757966f) * x2 + 1.89929986f));
    } else {
        return acos_positive_tail(x);
    }
}

// ---- cheap approximations used by light-tree importance, CoreRef.cpp:771-868 --------------------------
RT_HD float approx_atan2(float y, float x) {
    float t0, t1, t3, t4;
    t3 = fabsf(x);
    t1 = fabsf(y);
    t0 = fmaxf(t3, t1);
    t1 = fminf(t3, t1);
    t3 = 1.0f / t0;
    t3 = t1 * t3;
    t4 = t3 * t3;
    t0 = -0.013480470f;
    t0 = t0 * t4 + 0.057477314f;
    t0 = t0 * t4 - 0.121239071f;
    t0 = t0 * t4 + 0.195635925f;
    t0 = t0 * t4 - 0.332994597f;
    t0 = t0 * t4 + 0.999995630f;
    t3 = t0 * t3;
    t3 = (fabsf(y) > fabsf(x)) ? 1.570796327f - t3 : t3;
    t3 = (x < 0) ? 3.141592654f - t3 : t3;
    t3 = (y < 0) ? -t3 : t3;
    return t3;
}
RT_HD float approx_cos(float x) {
    const float tp = 1.0f / (2.0f * PI);
    x *= tp;
    x -= 0.25f + floorf(x + 0.25f);
    x *= 16.0f * (fabsf(x) - 0.5f);
    return x;
}
RT_HD float approx_acos(float x) {
    float negate = float(x < 0);
    x = fabsf(x);
    float ret = -0.0187293f;
    ret = ret * x;
    ret = ret + 0.0742610f;
    ret = ret * x;
    ret = ret - 0.2121144f;
    ret = ret * x;
    ret = ret + 1.5707288f;
    ret = ret * sqrtf(1.0f - x);
    ret = ret - 2 * negate * ret;
    return negate * PI + ret;
}

} // namespace rt
