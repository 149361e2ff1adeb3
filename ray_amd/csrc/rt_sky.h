// rt_sky.h -- the physical sky for narrow rays: what a camera ray or a mirror bounce sees when it leaves the scene and the
// environment is Ray::PhysicalSkyTexture.
//
// Reference: Ref::ShadeSky / ShadeSkyPrimary / ShadeSkySecondary (internal/AtmosphereRef.cpp:928-1024, called from
// RendererCPU.h:484-486, 555-557 over the rays ShadeSurface deferred, ShadeRef.cpp:1192-1196) and the integrator under them,
// Ref::IntegrateScattering (AtmosphereRef.cpp:606-889) with IntegrateScatteringMain (:480-595), the cloud density
// (:309-381), the texture and look-up-table samplers (:194-457) and the LUT parameterisation (:908-926).  Wide rays never come
// here: they read the environment map the host baked from the same integrator (SceneCPU.cpp:1017-1056), like any other map.
//
// What is restated and what is not.  The model is Hillaire's sky (single scattering marched along the view ray, transmittance
// and multiple-scattering look-up tables, the tables themselves computed by the reference's host code and handed over), a
// Schneider-style cloud layer (weather map x height gradient x eroding 3-d noise, curl-noise distortion, 48 view steps with a
// 24-step shadow march each, Wrenninge's three-octave phase approximation), a cirrus sheet, the sun disk, hashed stars and a
// textured moon.  The ORDER of the floating-point operations is the oracle's -- fvec4 arithmetic there is four-wide SSE, so
// four-component dot products add (x + y) + (w + z) and exp / pow run per component through libm -- because the host build of
// this file is compared with the reference bit for bit (tests/test_hostsim_parity.py: cornell_sky).  Names, structure and the
// decomposition into samplers / layers are this file's.
#pragma once

#include "shade_lights.h" // (the density of the importance sampling of the baked map: env_quadtree_pdf)

namespace rt {

// Constants.inl:149-160
constexpr int SKY_PRE_ATMOSPHERE_STEPS = 4, SKY_MAIN_ATMOSPHERE_STEPS = 12, SKY_CLOUD_STEPS = 48, SKY_CLOUD_SHADOW_STEPS = 24;
constexpr float SKY_CLOUDS_HORIZON_CUTOFF = 0.005f, SKY_CLOUDS_OFFSET_SCALE = 0.00007f, SKY_MOON_SUN_RELATION = 0.0000001f;
constexpr float SKY_STARS_THRESHOLD = 14.0f, SKY_SUN_BLEND_VAL = 0.000005f, SKY_SUN_MIN_ANGLE = 0.0005f;

// (SkyView -- rayhip_sky + the arrays it sizes -- is part of the scene view: rt_types.h)

RT_HD f4 sky_vec(const float v[4]) { return f4{v[0], v[1], v[2], v[3]}; }
RT_HD f4 sky_splat(const float v) { return f4{v, v, v, v}; }
RT_HD float sky_length(const f4 v) { return sqrtf(dot(v, v)); }
RT_HD f4 sky_fract(const f4 v) { return f4{fractf(v.x), fractf(v.y), fractf(v.z), fractf(v.w)}; }
RT_HD float sky_smoothstep(const float e0, const float e1, const float x) {
    const float t = saturatef((x - e0) / (e1 - e0));
    return t * t * (3.0f - 2.0f * t);
}
RT_HD float sky_remap(const float v, const float lo) { return saturatef((v - lo) / (1.000001f - lo)); }
RT_HD float sub_texel_uv(const float u, const float res) { return (u + 0.5f / res) * (res / (res + 1.0f)); } // Core.h:557
RT_HD float unit_float_from_bits(uint32_t m) { return uint_as_float((m & 0x007fffffu) | 0x3f800000u) - 1.0f; } // CoreRef.h:145
// the bit-twiddling exponential the reference uses for densities and step transmittances (AtmosphereRef.cpp:27-46)
RT_HD float sky_fast_exp(const float x) {
    int32_t i = int32_t(12102203.0f * x) + 127 * (1 << 23);
    const int32_t m = (i >> 7) & 0xffff;
    i += (((((((((((3537 * m) >> 16) + 13668) * m) >> 18) + 15817) * m) >> 14) - 80470) * m) >> 11);
    return int_as_float(i);
}
RT_HD f4 sky_fast_exp3(const f4 x) { return f4{sky_fast_exp(x.x), sky_fast_exp(x.y), sky_fast_exp(x.z), sky_fast_exp(0.0f)}; }

// ---- spheres around the planet's centre (0, -R, 0) -------------------------------------------------------------------------
// entry / exit parameters of the ray against a sphere, (-1, -1) if it misses
RT_HD f2 sky_sphere(f4 origin, const f4 dir, const f4 centre, const float radius) {
    origin = origin - centre;
    const float a = dot(dir, dir), b = 2.0f * dot(origin, dir), c = dot(origin, origin) - (radius * radius);
    float disc = b * b - 4 * a * c;
    if (disc < 0) {
        return f2{-1.0f, -1.0f};
    }
    disc = sqrtf(disc);
    return f2{(-b - disc) / (2 * a), (-b + disc) / (2 * a)};
}
RT_HD f4 planet_centre(const rayhip_atmosphere &at) { return f4{0.0f, -at.planet_radius, 0.0f, 0.0f}; }
RT_HD f2 sky_planet(const rayhip_atmosphere &at, const f4 o, const f4 d) { return sky_sphere(o, d, planet_centre(at), at.planet_radius); }
RT_HD f2 sky_atmosphere_shell(const rayhip_atmosphere &at, const f4 o, const f4 d) {
    return sky_sphere(o, d, planet_centre(at), at.planet_radius + at.atmosphere_height);
}
// (entry, exit) of the lower cloud shell in x, y and of the upper one in z, w
RT_HD f4 sky_cloud_shells(const rayhip_atmosphere &at, const f4 o, const f4 d) {
    const f2 lo = sky_sphere(o, d, planet_centre(at), at.planet_radius + at.clouds_height_beg);
    const f2 hi = sky_sphere(o, d, planet_centre(at), at.planet_radius + at.clouds_height_end);
    return f4{lo.x, lo.y, hi.x, hi.y};
}
// height above ground and the local "up"
RT_HD float sky_altitude(const rayhip_atmosphere &at, const f4 position, f4 &up) {
    up = position - planet_centre(at);
    const float r = sky_length(up);
    up = up / r;
    return r - at.planet_radius;
}

// ---- phase functions -----------------------------------------------------------------------------------------------------
RT_HD float phase_rayleigh(const float c) { return 3 * (1 + c * c) / (16 * PI); }
RT_HD float phase_mie(const float c) {
    const float g = fminf(0.85f, 0.9381f);
    const float k = 1.55f * g - 0.55f * g * g * g, kc = k * c;
    return (1 - k * k) / ((4 * PI) * (1 - kc) * (1 - kc));
}
RT_HD float phase_hg(const float mu, const float g) { return (1.0f - g * g) / (powf(1.0f + g * g - 2.0f * g * mu, 1.5f) * 4.0f * PI); }
RT_HD float phase_cloud(const float mu) { return mixf(phase_hg(mu, -0.2f), phase_hg(mu, 0.8f), 0.7f); }
// Wrenninge: three octaves of the cloud phase function at shrinking anisotropy
RT_HD f4 phase_cloud_octaves(const float mu) { return f4{phase_cloud(mu), phase_cloud(mu * 0.9f), phase_cloud(mu * 0.9f * 0.9f), 0.0f}; }
// energy reaching a cloud sample through optical thickness `towards_light`, three octaves weighted by their phase
RT_HD float cloud_light_energy(const float towards_light, const f4 phase) {
    const f4 curve = {expf(-towards_light * 0.8f), expf(-towards_light * 0.1f), expf(-towards_light * 0.002f), expf(-towards_light * 0.0f)};
    return dot(f4{2.0f, 0.8f, 0.4f, 0.0f} * phase, curve);
}

// ---- the participating medium at a height ---------------------------------------------------------------------------------
struct SkyMedium {
    f4 scattering, extinction, scattering_mie, scattering_rayleigh;
};
RT_HD SkyMedium sky_medium(const rayhip_atmosphere &at, const float h) {
    const float rayleigh = at.atmosphere_density * sky_fast_exp(-fmaxf(0.0f, h / at.rayleigh_height));
    const float mie = at.atmosphere_density * sky_fast_exp(-fmaxf(0.0f, h / at.mie_height));
    const float ozone = at.atmosphere_density * fmaxf(0.0f, 1.0f - fabsf(h - at.ozone_height_center) / at.ozone_half_width);
    SkyMedium m;
    m.scattering_mie = mie * sky_vec(at.mie_scattering);
    const f4 absorption_mie = mie * sky_vec(at.mie_absorption), extinction_mie = mie * sky_vec(at.mie_extinction);
    m.scattering_rayleigh = rayleigh * sky_vec(at.rayleigh_scattering);
    const f4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    const f4 extinction_rayleigh = m.scattering_rayleigh + zero;
    const f4 absorption_ozone = ozone * sky_vec(at.ozone_absorption);
    const f4 extinction_ozone = zero + absorption_ozone;
    m.scattering = m.scattering_mie + m.scattering_rayleigh + zero;
    (void)absorption_mie;
    m.extinction = extinction_mie + extinction_rayleigh + extinction_ozone;
    m.extinction.w = 1.0f; // (a safe divisor)
    return m;
}

// ---- samplers ---------------------------------------------------------------------------------------------------------------
RT_HD f4 sky_bilerp(const f4 t00, const f4 t01, const f4 t10, const f4 t11, const float kx, const float ky) {
    const f4 row0 = t01 * kx + t00 * (1.0f - kx), row1 = t11 * kx + t10 * (1.0f - kx);
    return row1 * ky + row0 * (1.0f - ky);
}
RT_HD f4 sky_lut_texel(const float *lut, const int w, const int x, const int y) {
    const float *p = lut + 4 * (y * w + x);
    return f4{p[0], p[1], p[2], p[3]};
}
RT_HD int sky_clampi(const int v, const int lo, const int hi) { return v < lo ? lo : (v > hi ? hi : v); }
RT_HD int sky_mini(const int a, const int b) { return a < b ? a : b; }
// transmittance towards the top of the atmosphere, by (cos of the zenith angle, height): clamped bilinear
RT_HD f4 sky_transmittance(const SkyView &s, f2 uv) {
    const int w = s.desc->transmittance_lut_w, h = s.desc->transmittance_lut_h;
    uv = f2{uv.x * float(w), uv.y * float(h)};
    const int x0 = sky_clampi(int(uv.x), 0, w - 1), y0 = sky_clampi(int(uv.y), 0, h - 1);
    const int x1 = sky_mini(x0 + 1, w - 1), y1 = sky_mini(y0 + 1, h - 1);
    return sky_bilerp(sky_lut_texel(s.transmittance_lut, w, x0, y0), sky_lut_texel(s.transmittance_lut, w, x1, y0),
                      sky_lut_texel(s.transmittance_lut, w, x0, y1), sky_lut_texel(s.transmittance_lut, w, x1, y1), fractf(uv.x), fractf(uv.y));
}
RT_HD f4 sky_multiscatter(const SkyView &s, f2 uv) {
    const int r = s.desc->multiscatter_lut_res;
    uv = f2{fractf(uv.x - 0.5f / float(r)) * float(r), fractf(uv.y - 0.5f / float(r)) * float(r)};
    const int x0 = sky_clampi(int(uv.x), 0, r - 1), y0 = sky_clampi(int(uv.y), 0, r - 1);
    const int x1 = sky_mini(x0 + 1, r - 1), y1 = sky_mini(y0 + 1, r - 1);
    return sky_bilerp(sky_lut_texel(s.multiscatter_lut, r, x0, y0), sky_lut_texel(s.multiscatter_lut, r, x1, y0),
                      sky_lut_texel(s.multiscatter_lut, r, x0, y1), sky_lut_texel(s.multiscatter_lut, r, x1, y1), fractf(uv.x), fractf(uv.y));
}
// LUT coordinates of (distance from the planet's centre, cos of the zenith angle): AtmosphereRef.cpp:908-926
RT_HD f2 sky_transmittance_uv(const rayhip_atmosphere &at, const float r, const float mu) {
    const float top = at.planet_radius + at.atmosphere_height;
    const float H = sqrtf(fmaxf(0.0f, top * top - at.planet_radius * at.planet_radius));
    const float rho = sqrtf(fmaxf(0.0f, r * r - at.planet_radius * at.planet_radius));
    const float disc = r * r * (mu * mu - 1.0f) + top * top;
    const float d = fmaxf(0.0f, (-r * mu + sqrtf(disc)));
    const float d_min = top - r, d_max = rho + H;
    return f2{(d - d_min) / (d_max - d_min), rho / H};
}
// the two look-ups a lit sample needs: transmittance towards `light`, and (optionally) the multiple-scattering term
RT_HD f4 sky_light_transmittance(const SkyView &s, const float altitude, const f4 up, const f4 light, f4 *multiscatter) {
    const rayhip_atmosphere &at = s.desc->atmosphere;
    const float mu = dot(light, up);
    const f4 tr = sky_transmittance(s, sky_transmittance_uv(at, altitude + at.planet_radius, mu));
    if (multiscatter) {
        *multiscatter = f4{0.0f, 0.0f, 0.0f, 0.0f};
        if (s.desc->multiscatter_lut_res != 0) {
            const float res = float(s.desc->multiscatter_lut_res);
            const f2 uv = {saturatef(mu * 0.5f + 0.5f), saturatef(altitude / at.atmosphere_height)};
            *multiscatter = sky_multiscatter(s, f2{sub_texel_uv(uv.x, res), sub_texel_uv(uv.y, res)});
        }
    }
    return tr;
}

// byte textures: `channels` interleaved bytes per texel, wrapped (power-of-two sizes)
RT_HD f4 sky_texel_rgb(const uint8_t *tex, const int w, const int x, const int y) {
    const uint8_t *p = tex + 3 * (y * w + x);
    return f4{float(p[0]), float(p[1]), float(p[2]), 0.0f};
}
RT_HD f4 sky_texel_rg(const uint8_t *tex, const int w, const int x, const int y) {
    const uint8_t *p = tex + 2 * (y * w + x);
    return f4{float(p[0]), float(p[1]), 0.0f, 0.0f};
}
// wrapped bilinear tap of an RGB8 map, texel centres at half-integers (weather, curl)
RT_HD f4 sky_tap_wrapped_rgb(const uint8_t *tex, const int res, const f2 uv_in) {
    const f2 uv = {fractf(uv_in.x - 0.5f / float(res)) * float(res), fractf(uv_in.y - 0.5f / float(res)) * float(res)};
    const int x0 = sky_clampi(int(uv.x), 0, res - 1), y0 = sky_clampi(int(uv.y), 0, res - 1);
    const int x1 = (x0 + 1) & (res - 1), y1 = (y0 + 1) & (res - 1);
    return sky_bilerp(sky_texel_rgb(tex, res, x0, y0), sky_texel_rgb(tex, res, x1, y0), sky_texel_rgb(tex, res, x0, y1), sky_texel_rgb(tex, res, x1, y1),
                      fractf(uv.x), fractf(uv.y)) * (1.0f / 255.0f);
}
RT_HD f4 sky_weather(const SkyView &s, const f2 uv) { return sky_tap_wrapped_rgb(s.weather, s.desc->weather_res, uv); }
RT_HD f4 sky_curl(const SkyView &s, const f2 uv) { return srgb_to_linear(sky_tap_wrapped_rgb(s.curl, s.desc->curl_res, uv)); }
// unwrapped coordinates, wrapped neighbours (moon: w x h; cirrus: two channels)
RT_HD f4 sky_moon_albedo(const SkyView &s, const f2 uv_in) {
    const int w = s.desc->moon_w, h = s.desc->moon_h;
    const f2 uv = {uv_in.x * float(w), uv_in.y * float(h)};
    const int x0 = sky_clampi(int(uv.x), 0, w - 1), y0 = sky_clampi(int(uv.y), 0, h - 1);
    const int x1 = (x0 + 1) & (w - 1), y1 = (y0 + 1) & (h - 1);
    return srgb_to_linear(sky_bilerp(sky_texel_rgb(s.moon, w, x0, y0), sky_texel_rgb(s.moon, w, x1, y0), sky_texel_rgb(s.moon, w, x0, y1),
                                     sky_texel_rgb(s.moon, w, x1, y1), fractf(uv.x), fractf(uv.y)) *
                          (1.0f / 255.0f));
}
RT_HD f4 sky_cirrus(const SkyView &s, const f2 uv_in) {
    const int r = s.desc->cirrus_res;
    const f2 uv = {uv_in.x * float(r), uv_in.y * float(r)};
    const int x0 = sky_clampi(int(uv.x), 0, r - 1), y0 = sky_clampi(int(uv.y), 0, r - 1);
    const int x1 = (x0 + 1) & (r - 1), y1 = (y0 + 1) & (r - 1);
    return srgb_to_linear(sky_bilerp(sky_texel_rg(s.cirrus, r, x0, y0), sky_texel_rg(s.cirrus, r, x1, y0), sky_texel_rg(s.cirrus, r, x0, y1),
                                     sky_texel_rg(s.cirrus, r, x1, y1), fractf(uv.x), fractf(uv.y)) *
                          (1.0f / 255.0f));
}
// wrapped trilinear tap of the 3-d noise volume, value in [0, 1]
RT_HD float sky_noise(const SkyView &s, const f4 uvw_in) {
    const int r = s.desc->noise3d_res;
    const float fr = float(r);
    const f4 uvw = {fractf(uvw_in.x - 0.5f / fr) * fr, fractf(uvw_in.y - 0.5f / fr) * fr, fractf(uvw_in.z - 0.5f / fr) * fr, 0.0f};
    const int x0 = sky_clampi(int(uvw.x), 0, r - 1), y0 = sky_clampi(int(uvw.y), 0, r - 1), z0 = sky_clampi(int(uvw.z), 0, r - 1);
    const int x1 = (x0 + 1) & (r - 1), y1 = (y0 + 1) & (r - 1), z1 = (z0 + 1) & (r - 1);
    const auto at = [&](const int x, const int y, const int z) { return float(s.noise3d[z * r * r + y * r + x]); };
    const float kx = fractf(uvw.x), ky = fractf(uvw.y), kz = fractf(uvw.z);
    const float n00 = (1.0f - kx) * at(x0, y0, z0) + kx * at(x1, y0, z0), n01 = (1.0f - kx) * at(x0, y1, z0) + kx * at(x1, y1, z0);
    const float n10 = (1.0f - kx) * at(x0, y0, z1) + kx * at(x1, y0, z1), n11 = (1.0f - kx) * at(x0, y1, z1) + kx * at(x1, y1, z1);
    const float n0 = (1.0f - ky) * n00 + ky * n01, n1 = (1.0f - ky) * n10 + ky * n11;
    return ((1.0f - kz) * n0 + kz * n1) / 255.0f;
}

// ---- the cloud layer --------------------------------------------------------------------------------------------------------------
// vertical profile of a cloud of `type` (0 stratus .. 0.5 stratocumulus .. 1 cumulus) at `height` (0 = base, 1 = top of the layer)
RT_HD float cloud_height_profile(const float height, const float type) {
    const float stratus = 1.0f - clampf(type * 2.0f, 0, 1), stratocumulus = 1.0f - fabsf(type - 0.5f) * 2.0f, cumulus = clampf(type - 0.5f, 0, 1) * 2.0f;
    const f4 g = f4{0.02f, 0.05f, 0.09f, 0.11f} * stratus + f4{0.02f, 0.2f, 0.48f, 0.625f} * stratocumulus + f4{0.01f, 0.0625f, 0.78f, 1.0f} * cumulus;
    return sky_smoothstep(g.x, g.y, height) - sky_smoothstep(g.z, g.w, height);
}
struct CloudSample {
    float density, height_fraction;
};
RT_HD CloudSample cloud_density(const SkyView &s, f4 position) {
    const rayhip_atmosphere &at = s.desc->atmosphere;
    f4 up;
    const float altitude = sky_altitude(at, position, up);
    CloudSample out;
    out.height_fraction = (altitude - at.clouds_height_beg) / (at.clouds_height_end - at.clouds_height_beg);
    out.density = 0.0f;
    const f4 weather = sky_weather(s, f2{SKY_CLOUDS_OFFSET_SCALE * (position.x + at.clouds_offset_x), SKY_CLOUDS_OFFSET_SCALE * (position.z + at.clouds_offset_z)});
    float coverage = mixf(weather.z, weather.y, at.clouds_variety);
    coverage = sky_remap(coverage, saturatef(1.0f - at.clouds_density + 0.5f * out.height_fraction));
    const float type = weather.x;
    coverage *= cloud_height_profile(out.height_fraction, type);
    if (out.height_fraction > 1.0f || coverage < 0.01f) {
        return out;
    }
    position = position / (1.5f * (at.clouds_height_end - at.clouds_height_beg));
    const f4 curl0 = sky_curl(s, f2{8.0f * position.x, 8.0f * position.z});
    position += curl0 * out.height_fraction * 0.25f;
    const f4 curl1 = sky_curl(s, f2{16.0f * position.y, 16.0f * position.x});
    position += f4{curl1.y, curl1.z, curl1.x, 0.0f} * (1.0f - out.height_fraction) * 0.05f;
    position += f4{at.clouds_flutter_x, 0.0f, at.clouds_flutter_z, 0.0f}; // (micro-movement)
    const float erosion = sky_noise(s, position);
    out.density = 3.0f * mixf(fmaxf(0.0f, 1.0f - type * 2.0f), 1.0f, out.height_fraction) * sky_remap(coverage, 0.6f * erosion);
    return out;
}
// optical thickness of the clouds between `from` and the light (a short march), 1 when the ray never meets the upper shell
RT_HD float cloud_shadow(const SkyView &s, const uint32_t rand_hash, const f4 from, const f4 towards) {
    if (!(sky_cloud_shells(s.desc->atmosphere, from, towards).w > 0)) {
        return 1.0f;
    }
    const float step = 16.0f;
    f4 p = from + unit_float_from_bits(rand_hash) * towards * step;
    float sum = 0.0f;
    for (int i = 0; i < SKY_CLOUD_SHADOW_STEPS; ++i) {
        sum += cloud_density(s, p).density;
        p += towards * step;
    }
    return sum * step;
}

// ---- stars -------------------------------------------------------------------------------------------------------------------------
RT_HD f4 star_gradient(const f4 cell) {
    const f4 h = {sinf(dot(cell, f4{127.1f, 311.7f, 74.7f, 0.0f})), sinf(dot(cell, f4{269.5f, 183.3f, 246.1f, 0.0f})),
                  sinf(dot(cell, f4{113.5f, 271.9f, 124.6f, 0.0f})), 0.0f};
    const f4 f = sky_fract(h * 43758.5453123f);
    return f4{-1.0f + 2.0f * f.x, -1.0f + 2.0f * f.y, -1.0f + 2.0f * f.z, -1.0f + 2.0f * f.w};
}
// gradient noise over the direction lattice
RT_HD float star_field(const f4 p) {
    const f4 i = {floorf(p.x), floorf(p.y), floorf(p.z), floorf(p.w)}, f = sky_fract(p);
    const f4 two_f = 2.0f * f;
    const f4 u = f * f * f4{3.0f - two_f.x, 3.0f - two_f.y, 3.0f - two_f.z, 3.0f - two_f.w};
    const auto corner = [&](const float cx, const float cy, const float cz) {
        const f4 c = {cx, cy, cz, 0.0f};
        return dot(star_gradient(i + c), f - c);
    };
    return mixf(mixf(mixf(corner(0, 0, 0), corner(1, 0, 0), u.x), mixf(corner(0, 1, 0), corner(1, 1, 0), u.x), u.y),
                mixf(mixf(corner(0, 0, 1), corner(1, 0, 1), u.x), mixf(corner(0, 1, 1), corner(1, 1, 1), u.x), u.y), u.z);
}

// ---- single scattering along a stretch of the view ray (IntegrateScatteringMain<false>, AtmosphereRef.cpp:480-595) ------------------
// returns the in-scattered radiance; `transmittance` is carried through
RT_HD f4 sky_march(const SkyView &s, const f4 start, const f4 dir, float length_, const f4 light_dir, const f4 moon_dir, const f4 light_color,
                   const float jitter, const int steps, f4 &transmittance) {
    const rayhip_atmosphere &at = s.desc->atmosphere;
    length_ = fminf(length_, sky_atmosphere_shell(at, start, dir).y);
    const f2 ground = sky_planet(at, start, dir);
    if (ground.x > 0) {
        length_ = fminf(length_, ground.x);
    }
    const float cos_sun = dot(dir, light_dir), cos_moon = dot(dir, moon_dir);
    const float sun_r = phase_rayleigh(cos_sun), sun_m = phase_mie(cos_sun), moon_r = phase_rayleigh(cos_moon), moon_m = phase_mie(cos_moon);

    f4 radiance = {0.0f, 0.0f, 0.0f, 0.0f};
    const float step = length_ / float(steps);
    float t = 0.1f * jitter * step;
    for (int i = 0; i < steps; ++i) {
        const f4 p = start + dir * t;
        f4 up;
        const float altitude = sky_altitude(at, p, up);
        const SkyMedium medium = sky_medium(at, altitude);
        const f4 depth = medium.extinction * step;
        const f4 through = sky_fast_exp3(f4{-depth.x, -depth.y, -depth.z, -depth.w});

        f4 in_scatter = {0.0f, 0.0f, 0.0f, 0.0f};
        if (light_dir.y > -0.025f) { // the sun
            f4 multi;
            const f4 tr = sky_light_transmittance(s, altitude, up, light_dir, &multi);
            const float unshadowed = sky_planet(at, p, light_dir).x > 0 ? 0.0f : 1.0f;
            const f4 phased = medium.scattering_rayleigh * sun_r + medium.scattering_mie * sun_m;
            in_scatter += (unshadowed * tr * phased + multi * medium.scattering) * light_color;
        } else if (at.moon_radius > 0.0f) { // moonlight: the sun's colour, scaled down
            f4 multi;
            const f4 tr = sky_light_transmittance(s, altitude, up, moon_dir, &multi);
            const f4 phased = medium.scattering_rayleigh * moon_r + medium.scattering_mie * moon_m;
            in_scatter += SKY_MOON_SUN_RELATION * (tr * phased + multi * medium.scattering) * light_color;
        }
        // (the reference also integrates the isotropic term of the multiple-scattering table here; its value is not used by this caller)
        const f4 integral = (in_scatter - in_scatter * through) / medium.extinction;
        radiance += transmittance * integral;
        transmittance *= through;
        t += step;
    }
    if (ground.x > 0) { // the ground, lit by the sun through the atmosphere
        f4 up;
        const float altitude = sky_altitude(at, start + dir * ground.x, up);
        const f4 tr = sky_light_transmittance(s, altitude, up, light_dir, nullptr);
        radiance += sky_vec(at.ground_albedo) * saturatef(dot(up, light_dir)) * transmittance * tr * light_color;
    }
    return radiance;
}

// ---- the view ray through the whole sky, for ONE light (IntegrateScattering, AtmosphereRef.cpp:606-889) ---------------------------------
RT_HD f4 sky_radiance_for_light(const SkyView &s, f4 start, const f4 dir, float length_, const f4 light_dir, const float light_angle, const f4 light_color,
                                const f4 light_color_point, uint32_t rand_hash) {
    const rayhip_atmosphere &at = s.desc->atmosphere;
    const f4 none = {0.0f, 0.0f, 0.0f, 0.0f};
    const f2 shell = sky_atmosphere_shell(at, start, dir);
    length_ = fminf(length_, shell.y);
    if (shell.x > 0) { // start at the entry point
        start += dir * shell.x;
        length_ -= shell.x;
    }
    const f2 ground = sky_planet(at, start, dir);
    if (ground.x > 0) {
        length_ = fminf(length_, ground.x);
    }
    if (length_ <= 0.0f) {
        return none;
    }
    const f2 moon_hit = sky_sphere(start, dir, sky_vec(at.moon_dir) * at.moon_distance, at.moon_radius);
    f4 moon_dir = sky_vec(at.moon_dir);
    {
        const f4 lit_point = moon_dir * at.moon_distance + 0.5f * light_dir * at.moon_radius;
        moon_dir = lit_point / sky_length(lit_point);
    }
    const float cos_sun = dot(dir, light_dir);
    const f4 sun_phase = phase_cloud_octaves(cos_sun), moon_phase = phase_cloud_octaves(dot(dir, moon_dir));
    const float brightness = light_color.x + light_color.y + light_color.z;

    f4 radiance = none, transmittance = {1.0f, 1.0f, 1.0f, 1.0f};
    const f4 shells = sky_cloud_shells(at, start, dir);
    const bool sky_above = ground.x < 0 && shells.y > 0 && brightness > 0.0f;

    // the air below the clouds
    if (shells.y > 0 && brightness > 0.0f) {
        const float jitter = unit_float_from_bits(rand_hash);
        rand_hash = hash(rand_hash);
        radiance += sky_march(s, start, dir, fminf(length_, shells.y), light_dir, moon_dir, light_color_point, jitter, SKY_PRE_ATMOSPHERE_STEPS, transmittance);
    }
    // the cloud layer
    if (sky_above && at.clouds_density > 0.0f && dir.y > SKY_CLOUDS_HORIZON_CUTOFF) {
        float span = fminf(length_, shells.w);
        const f4 entry = start + dir * shells.y;
        span -= shells.y;
        if (span > 0.0f) {
            const float step = span / float(SKY_CLOUD_STEPS);
            f4 p = entry + 1.0f * dir * unit_float_from_bits(rand_hash) * step;
            rand_hash = hash(rand_hash);
            // (transmittance towards the lights and the multiple-scattering term are taken once: the layer is ~500 m thick)
            f4 up, sun_multi, moon_multi;
            const float altitude = sky_altitude(at, p, up);
            const f4 sun_tr = sky_light_transmittance(s, altitude, up, light_dir, &sun_multi);
            const f4 moon_tr = sky_light_transmittance(s, altitude, up, moon_dir, &moon_multi);
            const f4 before = transmittance;
            f4 clouds = none;
            for (int i = 0; i < SKY_CLOUD_STEPS; ++i) {
                const CloudSample c = cloud_density(s, p);
                if (c.density > 0.0f) {
                    const float through = expf(-c.density * step);
                    const float ambient = (0.75f + 1.5f * fmaxf(0.0f, c.height_fraction - 0.1f));
                    if (light_dir.y > -0.025f) {
                        const float unshadowed = sky_planet(at, p, light_dir).x > 0 ? 0.0f : 1.0f;
                        const float thickness = cloud_shadow(s, rand_hash, p, light_dir);
                        clouds += transmittance * (sky_splat(unshadowed * cloud_light_energy(thickness, sun_phase)) + ambient * sun_multi) * (1.0f - through) * sun_tr;
                    } else if (at.moon_radius > 0.0f) {
                        const float thickness = cloud_shadow(s, rand_hash, p, moon_dir);
                        clouds += SKY_MOON_SUN_RELATION * transmittance * (sky_splat(cloud_light_energy(thickness, moon_phase)) + ambient * moon_multi) * (1.0f - through) * moon_tr;
                    }
                    transmittance *= through;
                    if (hsum(transmittance) < 0.01f) {
                        break;
                    }
                }
                p += dir * step;
            }
            // (towards the horizon the layer fades out: an arbitrary blend, as the reference says itself)
            float blend = saturatef((dir.y - (SKY_CLOUDS_HORIZON_CUTOFF + 0.25f)) / (SKY_CLOUDS_HORIZON_CUTOFF - (SKY_CLOUDS_HORIZON_CUTOFF + 0.25f)));
            blend = 1.0f - powf(blend, 5.0f);
            radiance += blend * clouds * light_color_point;
            transmittance = (1.0f - blend) * before + blend * transmittance;
        }
    }
    // the cirrus sheet
    if (sky_above && at.cirrus_clouds_amount > 0.0f) {
        f2 uv = {3e-4f * at.clouds_offset_z + 0.8f * dir.z / (fabsf(dir.y) + 0.02f), 3e-4f * at.clouds_offset_x + 0.8f * dir.x / (fabsf(dir.y) + 0.02f)};
        uv.y = uv.y + 1.75f;
        const auto amount = [&](const float noise_u, const float scale, const float gain) {
            float n = 1.0f - sky_noise(s, f4{fractf(noise_u), fractf(uv.x * scale), fractf(uv.y * scale), fractf(0.0f)});
            n = saturatef(n - 1.0f + at.cirrus_clouds_amount * gain) / (at.cirrus_clouds_amount + 1e-9f);
            return sky_smoothstep(0.0f, 1.0f, n);
        };
        float density = 1.2f * amount(0.0f, 0.03f, 0.6f) * sky_cirrus(s, f2{fractf(uv.x * 0.5f), fractf(uv.y * 0.5f)}).x;
        uv.x = uv.x + 0.25f;
        density += 0.6f * amount(0.7f, 0.02f, 0.7f) * sky_cirrus(s, f2{fractf(uv.x * 0.25f), fractf(uv.y * 0.25f)}).y;

        f4 up;
        const float altitude = sky_altitude(at, start + dir * at.cirrus_clouds_height, up);
        const f4 sun_tr = sky_light_transmittance(s, altitude, up, light_dir, nullptr), moon_tr = sky_light_transmittance(s, altitude, up, moon_dir, nullptr);
        if (light_dir.y > -0.025f) {
            radiance += transmittance * cloud_light_energy(0.002f, sun_phase) * sun_tr * density * light_color_point;
        } else if (at.moon_radius > 0.0f) {
            radiance += SKY_MOON_SUN_RELATION * transmittance * cloud_light_energy(0.002f, moon_phase) * moon_tr * density * light_color_point;
        }
        transmittance *= expf(-density * 0.002f * 1000.0f);
    }
    if (hsum(transmittance) < 0.001f) {
        return radiance;
    }
    // the air above the clouds
    if (ground.x < 0 && brightness > 0.0f) {
        const float jitter = unit_float_from_bits(rand_hash);
        rand_hash = hash(rand_hash);
        radiance += sky_march(s, start + dir * shells.w, dir, length_ - shells.y, light_dir, moon_dir, light_color_point, jitter, SKY_MAIN_ATMOSPHERE_STEPS, transmittance);
    }
    // the sun's disk
    if (light_angle > 0.0f && ground.x < 0.0f && brightness > 0.0f) {
        const float cos_edge = cosf(fmaxf(light_angle, SKY_SUN_MIN_ANGLE));
        const float soft = fminf(SKY_SUN_BLEND_VAL, 1.0f - cos_edge);
        radiance += transmittance * sky_smoothstep(cos_edge - soft, cos_edge + soft, cos_sun) * light_color;
    }
    // stars, where neither the ground nor the moon is in the way
    if (at.stars_brightness > 0.0f && ground.x < 0 && moon_hit.x < 0) {
        radiance += transmittance * (powf(clampf(star_field(dir * 400.0f), 0.0f, 1.0f), SKY_STARS_THRESHOLD) * at.stars_brightness);
    }
    // the moon
    if (ground.x < 0 && moon_hit.x > 0 && at.moon_radius > 0.0f && brightness > 0.0f) {
        const f4 centre = sky_vec(at.moon_dir) * at.moon_distance;
        f4 n = start + moon_hit.x * dir - centre;
        n = n / sky_length(n);
        const float theta = acosf(clampf(n.y, -1.0f, 1.0f)) / PI;
        float phi = atan2f(n.z, n.x);
        if (phi < 0) {
            phi += 2 * PI;
        }
        if (phi > 2 * PI) {
            phi -= 2 * PI;
        }
        radiance += transmittance * fmaxf(dot(n, light_dir), 0.0f) * sky_moon_albedo(s, f2{fractf(0.5f * phi / PI), theta});
    }
    return radiance;
}

// ---- one texel of the BAKED sky map (CalcSkyEnvTexture, SceneCommon.cpp:286-361; the reference's GPU scene runs the same per texel in a
// compute pass, SceneGPU.h:1697-1768): latitude-longitude direction of texel (x, y), the view-ray integral above summed over the directional
// lights (a light narrower than SKY_SUN_MIN_ANGLE is widened to it, its disk radiance rescaled), shared-exponent 8-bit encoding (rgb_to_rgbe,
// SceneCommon.cpp:7-17).  Same operations in the same order as the reference's host loop: the host build of this function is compared with the
// reference's baked map byte for byte (tests/test_sky_bake.py).
RT_HD uint32_t sky_rgbe8(const f4 rgb) {
    const float max_component = fmaxf(fmaxf(rgb.x, rgb.y), rgb.z);
    if (max_component < 1e-32) {
        return 0u;
    }
    int exponent;
    const float factor = frexpf(max_component, &exponent) * 256.0f / max_component;
    const f4 e = {rgb.x * factor, rgb.y * factor, rgb.z * factor, float(exponent + 128)};
    return uint32_t(uint8_t(e.x)) | (uint32_t(uint8_t(e.y)) << 8) | (uint32_t(uint8_t(e.z)) << 16) | (uint32_t(uint8_t(e.w)) << 24);
}
RT_HD uint32_t sky_bake_texel(const SkyView &s, const rayhip_light *lights, const int x, const int y, const int w, const int h) {
    const rayhip_atmosphere &at = s.desc->atmosphere;
    const float theta = PI * float(y) / float(h);
    const uint32_t px_hash = hash(uint32_t((x << 16) | y));
    const float phi = 2.0f * PI * (float(x) + 0.5f) / float(w);
    const f2 sincos_theta = portable_sincos(theta), sincos_phi = portable_sincos(phi);
    const f4 dir = {sincos_theta.x * sincos_phi.y, sincos_theta.y, sincos_theta.x * sincos_phi.x, 0.0f};
    const f4 eye = {0.0f, at.viewpoint_height, 0.0f, 0.0f};
    f4 color = {0.0f, 0.0f, 0.0f, 0.0f};
    if (s.dir_lights_count != 0) {
        for (uint32_t k = 0; k < s.dir_lights_count; ++k) {
            const rayhip_light &l = lights[s.dir_lights[k]];
            const f4 light_dir = {l.params[0], l.params[1], l.params[2], 0.0f};
            f4 light_col = {l.col[0], l.col[1], l.col[2], 0.0f}, point_col = light_col;
            if (l.params[4] != 0.0f) {
                const float radius = l.params[4];
                point_col *= (PI * radius * radius);
            }
            if (l.params[5] < SKY_SUN_MIN_ANGLE) {
                const float div = PI * tanf(SKY_SUN_MIN_ANGLE) * tanf(SKY_SUN_MIN_ANGLE);
                light_col = point_col / div;
            }
            color += sky_radiance_for_light(s, eye, dir, MAX_DIST, light_dir, l.params[5], light_col, point_col, px_hash);
        }
    } else if (at.stars_brightness > 0.0f) { // no sun: a stand-in below the horizon lights the moon
        const f4 light_dir = {0.0f, -1.0f, 0.0f, 0.0f}, light_col = {144809.859f, 129443.617f, 127098.89f, 0.0f};
        color += sky_radiance_for_light(s, eye, dir, MAX_DIST, light_dir, 0.0f, light_col, light_col, px_hash);
    }
    return sky_rgbe8(color);
}

// ---- what a deferred ray adds to its pixel (ShadeSky, AtmosphereRef.cpp:928-1010) ---------------------------------------------------
// `iteration`: the RenderScene iteration (the sky's random offsets are keyed by hash(iteration), not by the pass's rand_seed);
// `limit`: 3 x the direct / indirect clamp of the bounce, FLT_MAX without one
RT_HD f3 shade_sky_ray(const SceneView &sc, const Ray &ray, const Hit &hit, const int iteration, const int max_total_depth,
                       const float limit) {
    const SkyView &s = sc.sky;
    const uint32_t rand_hash = hash_combine(hash(ray.xy), hash(uint32_t(iteration)));
    const float rotation = is_indirect(ray.depth) ? sc.env.env_map_rotation : sc.env.back_map_rotation;
    const f4 dir = {ray.d.x * cosf(rotation) - ray.d.z * sinf(rotation), ray.d.y, ray.d.x * sinf(rotation) + ray.d.z * cosf(rotation), 0.0f};
    const float inv_pick_prob = (int(get_total_depth(ray.depth)) < max_total_depth) ? safe_div_pos(1.0f, hit.u) : -1.0f;
    const rayhip_atmosphere &at = s.desc->atmosphere;
    const f4 eye = {0.0f, at.viewpoint_height, 0.0f, 0.0f};

    f4 color = {0.0f, 0.0f, 0.0f, 0.0f};
    if (s.dir_lights_count != 0) {
        for (uint32_t k = 0; k < s.dir_lights_count; ++k) {
            const rayhip_light &l = sc.lights[s.dir_lights[k]];
            const f4 light_dir = {l.params[0], l.params[1], l.params[2], 0.0f}, light_col = {l.col[0], l.col[1], l.col[2], 0.0f};
            f4 point_col = light_col;
            if (l.params[4] != 0.0f) { // tan of the angular radius: the disk's radiance -> the irradiance of a point light
                const float radius = l.params[4];
                point_col *= (PI * radius * radius);
            }
            color += sky_radiance_for_light(s, eye, dir, MAX_DIST, light_dir, l.params[5], light_col, point_col, rand_hash);
        }
    } else if (at.stars_brightness > 0.0f) { // no sun: a stand-in below the horizon lights the moon
        const f4 light_dir = {0.0f, -1.0f, 0.0f, 0.0f}, light_col = {144809.859f, 129443.617f, 127098.89f, 0.0f};
        color += sky_radiance_for_light(s, eye, dir, MAX_DIST, light_dir, 0.0f, light_col, light_col, rand_hash);
    }
    if (sc.env.light_index != 0xffffffff && inv_pick_prob >= 0.0f && is_indirect(ray.depth)) { // MIS against the sampling of the baked map
        const float light_pdf = sc.env.qtree_levels ? safe_div_pos(env_quadtree_pdf(sc, rotation, f3{dir.x, dir.y, dir.z}), inv_pick_prob)
                                                    : safe_div_pos(0.5f, PI * inv_pick_prob);
        color *= power_heuristic(ray.pdf, light_pdf);
    }
    color *= f4{ray.c.x, ray.c.y, ray.c.z, 0.0f};
    const float sum = hsum(color);
    if (sum > limit) {
        color *= (limit / sum);
    }
    return f3{color.x, color.y, color.z};
}

} // namespace rt
