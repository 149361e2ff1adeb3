// rt_base.h -- shared constants, small vector types and scalar helpers for the HIP path-tracer kernels.
//
// Every kernel body in this directory is written as an RT_HD (host+device) inline function over plain
// structs so that (a) hipcc compiles it for gfx950 inside kernels.hip and (b) tests/hostsim can compile the
// very same source with g++ and step through it next to the reference on a box without a GPU.  The
// product library never contains a host path: kernels.hip is the only translation unit that ships.
//
// Arithmetic order is part of the contract.  The reference's scalar backend does its vector math with a
// 4-lane SSE2 type (internal/simd/simd_sse.h) built WITHOUT fma (CMakeLists.txt:47: -msse2 -mno-avx), and
// a path tracer is chaotic in its inputs, so the helpers below reproduce its association order:
//   dot(a,b)    = (a0*b0 + a1*b1) + (a3*b3 + a2*b2)            simd_sse.h:252-260
//   length(a)   = sqrt(dot(a,a)),  normalize(a) = a / length(a) (a true divide per lane)  simd_sse.h:120-130,288
//   hsum(a)     = ((a0 + a1) + a2) + a3                          simd_sse.h:144-152 (non-SSE4.1 branch)
// f3 is used where lane 3 is known to be zero in the reference (then a3*b3 = +0 and the sum is unchanged
// up to the sign of a zero result); f4 is used where lane 3 carries data.
// Build with -ffp-contract=off for bit-parity with the reference; see DESIGN.md "Numerics".
#pragma once

// RT_PROF(k): section marker of the shade-kernel profile build (tools/variants.py, -DRT_PROFILE_SHADE); nothing otherwise
#ifndef RT_PROF
#define RT_PROF(k)
#endif
// RT_PROF_SHADE_LANES(k): lane census of the shade-kernel profile build (shade_kernels.hip, -DRT_PROFILE_SHADE): active lanes into
// slot k, wave-level executions into slot k + 1
#ifndef RT_PROF_SHADE_LANES
#define RT_PROF_SHADE_LANES(k)
#endif
// same for the traversal-kernel profile build (-DRT_PROFILE_TRACE); RT_PROF_WAIT forces the loads just issued to land
// before the next marker so that "waiting for memory" becomes its own section
#ifndef RT_PROF_T
#define RT_PROF_T(k)
#define RT_PROF_WAIT(...)
#define RT_PROF_LANES(k) // counts active lanes into slot k and wave-level executions into slot k+1
#endif

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#define RT_D __device__ __forceinline__
#else
#include <cmath>
#include <cstring>
#define RT_HD inline __attribute__((always_inline))
#define RT_D inline
#endif

// RT_HD_RARE: bodies that run for a few rays in a thousand (a transparent surface was crossed) and whose registers should
// not be charged to the loop around them: a real call on the device when RT_NOINLINE_RARE is set (tuning)
#if defined(__HIPCC__) && defined(RT_NOINLINE_RARE)
#define RT_HD_RARE __host__ __device__ __attribute__((noinline))
#else
#define RT_HD_RARE RT_HD
#endif

#include <float.h>
#include <math.h>

namespace rt {

// ---- constants: reference internal/Constants.inl --------------------------------------------------
constexpr int MAX_STACK_SIZE = 48;
constexpr float HIT_BIAS = 0.00001f;
constexpr float HIT_EPS = 0.000001f;
constexpr float FLT_EPS_ = 0.0000001f;
constexpr float MAX_DIST = 3.402823466e+30F;
constexpr float SPHERICAL_AREA_THRESHOLD = 0.00005f;
constexpr uint32_t LEAF_NODE_BIT = (1u << 31);
constexpr uint32_t PRIM_INDEX_BITS = ~LEAF_NODE_BIT;
constexpr uint32_t BVH2_PRIM_COUNT_BITS = (7u << 29);
constexpr uint32_t BVH2_PRIM_INDEX_BITS = ~BVH2_PRIM_COUNT_BITS;
constexpr float PI = 3.141592653589793238463f;

constexpr int RAND_DIM_FILTER = 0;
constexpr int RAND_DIM_LENS = 1;
constexpr int RAND_DIM_BASE_COUNT = 2;
constexpr int RAND_DIM_BSDF_PICK = 0;
constexpr int RAND_DIM_BSDF = 1;
constexpr int RAND_DIM_LIGHT_PICK = 2;
constexpr int RAND_DIM_LIGHT = 3;
constexpr int RAND_DIM_TEX = 4;
constexpr int RAND_DIM_CACHE = 5;
constexpr int RAND_DIM_BOUNCE_COUNT = 8;
constexpr int RAND_SAMPLES_COUNT = 4096; // __pmj02_sample_count
constexpr int RAND_DIMS_COUNT = 32;      // __pmj02_dims_count

constexpr int LIGHT_TYPE_SPHERE = 0;
constexpr int LIGHT_TYPE_DIR = 1;
constexpr int LIGHT_TYPE_LINE = 2;
constexpr int LIGHT_TYPE_RECT = 3;
constexpr int LIGHT_TYPE_DISK = 4;
constexpr int LIGHT_TYPE_TRI = 5;
constexpr int LIGHT_TYPE_ENV = 6;

constexpr int RAY_TYPE_CAMERA = 0;
constexpr int RAY_TYPE_DIFFUSE = 1;
constexpr int RAY_TYPE_SPECULAR = 2;
constexpr int RAY_TYPE_REFR = 3;
constexpr int RAY_TYPE_SHADOW = 4;
constexpr uint32_t RAY_TYPE_DIFFUSE_BIT = (1u << RAY_TYPE_DIFFUSE);
constexpr uint32_t RAY_TYPE_SPECULAR_BIT = (1u << RAY_TYPE_SPECULAR);
constexpr uint32_t RAY_TYPE_REFR_BIT = (1u << RAY_TYPE_REFR);

constexpr int NORMALS_TEXTURE = 0;
constexpr int BASE_TEXTURE = 1;
constexpr int ROUGH_TEXTURE = 2;
constexpr int METALLIC_TEXTURE = 3;
constexpr int SPECULAR_TEXTURE = 4;
constexpr int MIX_MAT1 = 3;
constexpr int MIX_MAT2 = 4;
constexpr int MATERIAL_SOLID_BIT = 32768;
constexpr int MATERIAL_INDEX_BITS = 16383;
constexpr uint32_t MAT_FLAG_IMP_SAMPLE = (1u << 0u);
constexpr uint32_t MAT_FLAG_MIX_ADD = (1u << 1u);
constexpr int MAX_MIP_LEVEL = 11;
constexpr float MAX_CONE_SPREAD_INCREMENT = 0.05f;
constexpr int FILTER_TABLE_SIZE = 1024;

// Core.h:162-164
constexpr uint32_t TEX_SRGB_BIT = (1u << 24);
constexpr uint32_t TEX_RECONSTRUCT_Z_BIT = (2u << 24);
constexpr uint32_t TEX_YCOCG_BIT = (4u << 24);

// Ray::eShadingNode, SceneBase.h:46
enum : uint32_t { NODE_DIFFUSE = 0, NODE_GLOSSY, NODE_REFRACTIVE, NODE_EMISSIVE, NODE_MIX, NODE_TRANSPARENT, NODE_PRINCIPLED };

// ---- bit casts ------------------------------------------------------------------------------------
RT_HD int32_t float_as_int(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_int(v);
#else
    int32_t i;
    memcpy(&i, &v, 4);
    return i;
#endif
}
RT_HD float int_as_float(int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __int_as_float(v);
#else
    float f;
    memcpy(&f, &v, 4);
    return f;
#endif
}
RT_HD uint32_t float_as_uint(float v) { return uint32_t(float_as_int(v)); }
RT_HD float uint_as_float(uint32_t v) { return int_as_float(int32_t(v)); }

// ---- vector types ---------------------------------------------------------------------------------
struct f2 {
    float x, y;
};
struct f3 {
    float x, y, z;
};
struct f4 {
    float x, y, z, w;
};

RT_HD f2 mk2(float x, float y) { return f2{x, y}; }
RT_HD f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
RT_HD f3 mk3(const float *p) { return f3{p[0], p[1], p[2]}; }
RT_HD f3 splat3(float v) { return f3{v, v, v}; }
RT_HD f4 mk4(float x, float y, float z, float w) { return f4{x, y, z, w}; }
RT_HD f4 mk4(const f3 &v, float w) { return f4{v.x, v.y, v.z, w}; }
RT_HD f3 xyz(const f4 &v) { return f3{v.x, v.y, v.z}; }

RT_HD f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
RT_HD f2 operator-(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }
RT_HD f2 operator*(f2 a, f2 b) { return {a.x * b.x, a.y * b.y}; }
RT_HD f2 operator*(f2 a, float b) { return {a.x * b, a.y * b}; }
RT_HD f2 operator*(float a, f2 b) { return {a * b.x, a * b.y}; }

RT_HD f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_HD f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_HD f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_HD f3 operator/(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
RT_HD f3 operator*(f3 a, float b) { return {a.x * b, a.y * b, a.z * b}; }
RT_HD f3 operator*(float a, f3 b) { return {a * b.x, a * b.y, a * b.z}; }
RT_HD f3 operator/(f3 a, float b) { return {a.x / b, a.y / b, a.z / b}; }
RT_HD f3 operator+(f3 a, float b) { return {a.x + b, a.y + b, a.z + b}; }
RT_HD f3 operator-(f3 a, float b) { return {a.x - b, a.y - b, a.z - b}; }
RT_HD f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
RT_HD f3 &operator+=(f3 &a, f3 b) { return a = a + b; }
RT_HD f3 &operator-=(f3 &a, f3 b) { return a = a - b; }
RT_HD f3 &operator*=(f3 &a, f3 b) { return a = a * b; }
RT_HD f3 &operator*=(f3 &a, float b) { return a = a * b; }
RT_HD f3 &operator/=(f3 &a, float b) { return a = a / b; }

RT_HD f4 operator+(f4 a, f4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
RT_HD f4 operator-(f4 a, f4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
RT_HD f4 operator*(f4 a, f4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
RT_HD f4 operator/(f4 a, f4 b) { return {a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w}; }
RT_HD f4 operator*(f4 a, float b) { return {a.x * b, a.y * b, a.z * b, a.w * b}; }
RT_HD f4 operator*(float a, f4 b) { return {a * b.x, a * b.y, a * b.z, a * b.w}; }
RT_HD f4 operator/(f4 a, float b) { return {a.x / b, a.y / b, a.z / b, a.w / b}; }
RT_HD f4 &operator+=(f4 &a, f4 b) { return a = a + b; }
RT_HD f4 &operator*=(f4 &a, f4 b) { return a = a * b; }
RT_HD f4 &operator*=(f4 &a, float b) { return a = a * b; }

// simd_sse.h:252-260 with lane 3 == 0 on at least one side
RT_HD float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// full 4-lane form
RT_HD float dot(f4 a, f4 b) { return (a.x * b.x + a.y * b.y) + (a.w * b.w + a.z * b.z); }
// generic (non-SSE) fixed_size_simd<float,2>::dot, simd.h:475-479: ((0 + a0*b0) + a1*b1)
RT_HD float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
RT_HD float length2(f3 a) { return dot(a, a); }
RT_HD float length(f3 a) { return sqrtf(dot(a, a)); }
RT_HD float length(f2 a) { return sqrtf(a.x * a.x + a.y * a.y); }
RT_HD float length2(f2 a) { return a.x * a.x + a.y * a.y; }
RT_HD f3 normalize(f3 a) { return a / length(a); }
RT_HD f3 normalize_len(f3 a, float &out_len) {
    out_len = length(a);
    return a / out_len;
}
// CoreRef.h:281-285
RT_HD f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// simd_sse.h:144-152, SSE2 branch: comp[0]+comp[1]+comp[2]+comp[3]
RT_HD float hsum(f4 a) { return ((a.x + a.y) + a.z) + a.w; }

// _mm_min_ps / _mm_max_ps return the SECOND operand when either is NaN or both are equal
RT_HD float sse_min(float a, float b) { return a < b ? a : b; }
RT_HD float sse_max(float a, float b) { return a > b ? a : b; }

// ---- scalar helpers: CoreRef.h / Core.h -----------------------------------------------------------
RT_HD float clampf(float val, float mn, float mx) { return val < mn ? mn : (val > mx ? mx : val); } // Core.h:537
RT_HD int clampi(int val, int mn, int mx) { return val < mn ? mn : (val > mx ? mx : val); }
RT_HD float saturatef(float v) { return clampf(v, 0.0f, 1.0f); }
RT_HD float sqr(float x) { return x * x; }
RT_HD float mixf(float v1, float v2, float k) { return (1.0f - k) * v1 + k * v2; } // ShadeRef.cpp:30
RT_HD f3 mix3(f3 v1, f3 v2, float k) { return (1.0f - k) * v1 + k * v2; }        // simd.h:609-612
RT_HD float fractf(float v) { return v - floorf(v); }                              // CoreRef.h:152
RT_HD float safe_sqrt(float v) { return sqrtf(fmaxf(v, 0.0f)); }                   // CoreRef.h:154
RT_HD float safe_div(float a, float b) { return b != 0.0f ? (a / b) : FLT_MAX; }   // CoreRef.h:162
RT_HD float safe_div_pos(float a, float b) { return a / fmaxf(b, FLT_EPS_); }      // CoreRef.h:170
RT_HD float safe_div_neg(float a, float b) { return a / fminf(b, -FLT_EPS_); }     // CoreRef.h:178
// CoreRef.h:186-190
RT_HD f3 safe_invert(f3 v) {
    f3 r;
    r.x = 1.0f / ((fabsf(v.x) > FLT_EPS_) ? v.x : copysignf(FLT_EPS_, v.x));
    r.y = 1.0f / ((fabsf(v.y) > FLT_EPS_) ? v.y : copysignf(FLT_EPS_, v.y));
    r.z = 1.0f / ((fabsf(v.z) > FLT_EPS_) ? v.z : copysignf(FLT_EPS_, v.z));
    return r;
}
// CoreRef.h:192-199
RT_HD f3 safe_normalize(f3 a) {
    const float l = length(a);
    return l > 0.0f ? (a / l) : a;
}
RT_HD float lum(f3 c) { return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z; } // CoreRef.h:398
RT_HD float power_heuristic(float a, float b) {                                        // CoreRef.h:424
    const float t = a * a;
    return t / (b * b + t);
}
// CoreRef.h:406-417
RT_HD float fast_log2(float val) {
    int32_t x = float_as_int(val);
    float log_2 = float(((x >> 23) & 255) - 128);
    x &= ~(255 << 23);
    x += 127 << 23;
    const float v = int_as_float(x);
    log_2 += ((-0.34484843f) * v + 2.02466578f) * v - 0.67487759f;
    return log_2;
}
// CoreRef.h:447-462 "A Fast and Robust Method for Avoiding Self-Intersection"
RT_HD f3 offset_ray(f3 p, f3 n) {
    const float Origin = 1.0f / 32.0f;
    const float FloatScale = 1.0f / 65536.0f;
    const float IntScale = 128.0f;
    // ivec4(IntScale * n): _mm_cvttps_epi32 truncation
    const int ox = int(IntScale * n.x), oy = int(IntScale * n.y), oz = int(IntScale * n.z);
    const f3 p_i = {int_as_float(float_as_int(p.x) + ((p.x < 0.0f) ? -ox : ox)),
                    int_as_float(float_as_int(p.y) + ((p.y < 0.0f) ? -oy : oy)),
                    int_as_float(float_as_int(p.z) + ((p.z < 0.0f) ? -oz : oz))};
    return f3{fabsf(p.x) < Origin ? (p.x + FloatScale * n.x) : p_i.x,
              fabsf(p.y) < Origin ? (p.y + FloatScale * n.y) : p_i.y,
              fabsf(p.z) < Origin ? (p.z + FloatScale * n.z) : p_i.z};
}

// ---- ray depth packing: CoreRef.h:253-280 ----------------------------------------------------------
RT_HD uint32_t mask_ray_depth(uint32_t depth) { return depth & 0x0fffffff; }
RT_HD uint32_t pack_ray_type(int ray_type) { return uint32_t(ray_type << 28); }
RT_HD uint32_t pack_ray_depth(int diff, int spec, int refr, int transp) {
    return uint32_t((diff << 0) | (spec << 7) | (refr << 14) | (transp << 21));
}
RT_HD int get_diff_depth(uint32_t depth) { return int(depth & 0x7f); }
RT_HD int get_spec_depth(uint32_t depth) { return int(depth >> 7) & 0x7f; }
RT_HD int get_refr_depth(uint32_t depth) { return int(depth >> 14) & 0x7f; }
RT_HD int get_transp_depth(uint32_t depth) { return int(depth >> 21) & 0x7f; }
RT_HD int get_total_depth(uint32_t depth) {
    return get_diff_depth(depth) + get_spec_depth(depth) + get_refr_depth(depth) + get_transp_depth(depth);
}
RT_HD int get_ray_type(uint32_t depth) { return int(depth >> 28) & 0xf; }
RT_HD bool is_indirect(uint32_t depth) { return (depth & 0x001fffff) != 0; }

// ---- transforms: CoreRef.cpp:2789-2816 --------------------------------------------------------------
RT_HD f3 transform_point(f3 p, const float *m) {
    return f3{m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
              m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
}
RT_HD f3 transform_direction(f3 p, const float *m) {
    return f3{m[0] * p.x + m[4] * p.y + m[8] * p.z, m[1] * p.x + m[5] * p.y + m[9] * p.z,
              m[2] * p.x + m[6] * p.y + m[10] * p.z};
}
RT_HD f3 transform_normal(f3 n, const float *im) {
    return f3{im[0] * n.x + im[1] * n.y + im[2] * n.z, im[4] * n.x + im[5] * n.y + im[6] * n.z,
              im[8] * n.x + im[9] * n.y + im[10] * n.z};
}
RT_HD f3 world_from_tangent(f3 T, f3 B, f3 N, f3 V) { return V.x * T + V.y * B + V.z * N; } // CoreRef.h:296
RT_HD f3 tangent_from_world(f3 T, f3 B, f3 N, f3 V) { return f3{dot(V, T), dot(V, B), dot(V, N)}; }

// ---- multi-GPU tile sharding ---------------------------------------------------------
// The frame is cut into `tile` x `tile` squares walked row-major and dealt round-robin to `count` ranks; every
// rank owns the pixels of its tiles for the whole render.  Pixels are independent (RNG is keyed by x,y,iteration:
// CoreRef.cpp:1477-1478), so the union of the ranks' images is bit-identical to a single-GPU render.
// Iteration batching.  Several iterations (samples per pixel) of the same rect can be in flight in ONE wavefront pass:
// iteration first + l lives on "layer" l of a virtual frame that is `cols` real frames wide and ceil(count / cols) high:
// layer l sits at column l % cols, row l / cols, i.e. a ray of pixel (x, y) on layer l carries
// xy = ((x + (l % cols) * frame_w) << 16) | (y + (l / cols) * frame_h) -- both halves of ray_data_t::xy are 16 bits, which
// bounds a pass to (65535 / frame_w) * (65535 / frame_h) layers -- and every per-iteration buffer (temp, primary aux
// outputs) is one such virtual frame.  Whatever depends on the iteration (sample index, rand_seed, running-mean weights) is derived
// per ray from its layer, the random-number hash from the REAL pixel, and k_accumulate folds the layers into the
// running means in iteration order -- so a batch is bit-identical to the same iterations rendered one by one, but a
// launch carries layers x more rays: small frames (a GPU's share of a tile-sharded frame, 256x256 previews) fill the
// machine and the fixed cost of a launch (tail of its longest rays) is paid once per batch.
struct Layering {
    int frame_h; // height of the real frame
    int count;   // layers in this pass (1 = plain single-iteration pass)
    int frame_w; // width of the real frame
    int cols;    // layers side by side in the virtual frame (>= 1)
};
RT_HD Layering single_layer(const int w, const int h) { return Layering{h, 1, w, 1}; }
// offset of layer `layer` inside the virtual frame, in the packed form of ray_data_t::xy
RT_HD uint32_t layer_offset_xy(const Layering L, const uint32_t layer) {
    const uint32_t lx = layer % uint32_t(L.cols), ly = layer / uint32_t(L.cols);
    return ((lx * uint32_t(L.frame_w)) << 16) + ly * uint32_t(L.frame_h);
}
RT_HD uint32_t xy_layer(const uint32_t xy, const Layering L) {
    return L.count > 1 ? ((xy & 0xffffu) / uint32_t(L.frame_h)) * uint32_t(L.cols) + (xy >> 16) / uint32_t(L.frame_w) : 0u;
}
RT_HD uint32_t xy_real(const uint32_t xy, const Layering L, const uint32_t layer) { return xy - layer_offset_xy(L, layer); }
// width of the virtual frame = row pitch of the per-iteration buffers
RT_HD int virtual_width(const Layering L) { return L.frame_w * L.cols; }

struct Shard {
    int tile, count, index;
};
RT_HD bool pixel_owned(const Shard &sh, const int img_w, const int x, const int y) {
    if (sh.count <= 1) {
        return true;
    }
    const int tiles_x = (img_w + sh.tile - 1) / sh.tile;
    return ((y / sh.tile) * tiles_x + (x / sh.tile)) % sh.count == sh.index;
}

// The frame exchange of a tile-sharded render packs the tiles a rank owns DENSELY: owned tile j of rank r is frame tile
// r + j * N (row-major over the frame's shard tiles), one tile x tile slot each (ragged edge tiles keep their full slot).
struct ShardTiles {
    int tiles_x, total; // shard tiles per frame row, in the frame
};
RT_HD ShardTiles shard_tiles(const int w, const int h, const int tile) {
    ShardTiles t;
    t.tiles_x = (w + tile - 1) / tile;
    t.total = t.tiles_x * ((h + tile - 1) / tile);
    return t;
}
// tiles rank `index` of `count` owns
RT_HD int shard_owned_tiles(const int total, const int count, const int index) {
    return index < total ? (total - index + count - 1) / count : 0;
}
// slot `i` of the packed buffer of shard `sh` -> its pixel; false for the padding of a ragged tile
RT_HD bool shard_slot_pixel(const Shard &sh, const int w, const int h, const int i, int &x, int &y) {
    const int per_tile = sh.tile * sh.tile, tiles_x = (w + sh.tile - 1) / sh.tile;
    const int t = sh.index + (i / per_tile) * sh.count, l = i % per_tile;
    x = (t % tiles_x) * sh.tile + l % sh.tile, y = (t / tiles_x) * sh.tile + l / sh.tile;
    return x < w && y < h;
}

} // namespace rt
