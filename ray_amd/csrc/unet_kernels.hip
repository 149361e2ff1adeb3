// unet_kernels.hip -- one 3 x 3 convolution pass of the UNet denoiser as an implicit GEMM on the f32 matrix cores (unet.h).
//
// Workgroup = 4 wavefronts = an 8 x 16 tile of output pixels x ALL output channels.  The K dimension (9 taps x input
// channels) is walked in chunks of 16 channels: the chunk's (8 + 2) x (16 + 2) input patch and its 9 x 16 x N weight rows
// are staged in LDS (patch pixels 20 floats apart, weight rows at a pitch that spreads the four k-rows of a fetch over the
// four 16-bank groups: both operand fetches of a wavefront are conflict-free; since round 4 patch pixels are 20 floats apart and the
// staging is software-pipelined, see k_conv3x3), then every wavefront issues, per tap and per
// 4 channels, one A fetch per pixel row and one B fetch per 16 output channels and 2 x n_tiles v_mfma_f32_16x16x4_f32:
//   A operand  lane l: pixel m = l & 15 of the row, channel k = l >> 4        (16 pixels x 4 channels)
//   B operand  lane l: out channel n = l & 15,      channel k = l >> 4        (4 channels x 16 out channels)
//   D          lane l: out channel n = l & 15, pixels 4 (l >> 4) + 0 .. 3     (4 accumulator registers)
// The pre-operations of the reference's kernels are applied while the patch is staged (nearest-neighbour upsample of the
// first input incl. its zero border, Convolution.h:344-352; HDR transfer / positive-normalise of the image inputs with zeros
// outside the image, :222-244), the post-operations in the epilogue (bias is the accumulator's start value, ReLU, 2 x 2 max
// pool :180-191, inverse HDR transfer :98-113).
#include "unet.h"

namespace rt {
namespace unet {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int PATCH_W = TILE_W + 2;
constexpr int PATCH_PITCH = CHUNK + 4; // floats per staged pixel: 16-byte aligned, and 20 m mod 64 spreads the 16 pixels of an A fetch over the banks

// Convolution.h:66-113 (namespace transfer)
__device__ __forceinline__ float transfer_in_hdr(const float val) {
    const float a = 1.41283765e+03f, b = 1.64593172e+00f, c = 4.31384981e-01f, d = -2.94139609e-03f, e = 1.92653254e-01f, f = 6.26026094e-03f,
                g = 9.98620152e-01f, y0 = 1.57945760e-06f, y1 = 3.22087631e-02f, norm_scale = 0.318967164f;
    if (val <= y0) {
        return a * val * norm_scale;
    } else if (val <= y1) {
        return (b * powf(val, c) + d) * norm_scale;
    }
    return (e * logf(val + f) + g) * norm_scale;
}
__device__ __forceinline__ float transfer_out_hdr(float val) {
    const float a = 1.41283765e+03f, b = 1.64593172e+00f, c = 4.31384981e-01f, d = -2.94139609e-03f, e = 1.92653254e-01f, f = 6.26026094e-03f,
                g = 9.98620152e-01f, x0 = 2.23151711e-03f, x1 = 3.70974749e-01f, norm_scale = 3.13511896f;
    val *= norm_scale;
    if (val <= x0) {
        return val / a;
    } else if (val <= x1) {
        return powf((val - d) / b, 1.0f / c);
    }
    return expf((val - g) / e) - f;
}

// Staging, round 4.  Round 3 staged every chunk element by element (a division chain and a bounds test per float, the global load
// consumed by the very next LDS store) between two barriers, with the matrix cores idle: 43 % of their f32 peak.  Now
//   * a thread moves 16-byte pieces: four channels of one patch pixel (NHWC: contiguous), four consecutive weights; which pixels a
//     thread serves, where they live in the inputs and in LDS, and whether they are inside the tensor is worked out ONCE, before the
//     chunk loop (the patch geometry does not depend on the chunk);
//   * the global loads of chunk c + 1 are issued BEFORE the matrix instructions of chunk c and land in registers while those run; only
//     the short register -> LDS copy stands between the two barriers of a chunk (software pipelining through registers: the LDS chunk
//     stays single, so the occupancy does);
//   * patch pixels are 20 floats apart (16-byte aligned for ds_write_b128; 20 m mod 64 is a different multiple of four for each of the 16
//     pixels of an A fetch, so the four channels of the four k-lanes still land in 64 different banks).
// The nine image channels of the first and of the third-last pass are a tensor like any other: k_image_inputs below writes them once.
// ROWS pixel rows per wavefront (a workgroup's tile is 4 ROWS x 16 pixels): every B fetch feeds ROWS matrix instructions and a chunk's
// weights are staged once per 4 ROWS x 16 pixels -- 4 where the accumulators (4 ROWS NT registers) and the LDS footprint leave the
// occupancy where it was (NT <= 4: the full- and half-resolution passes, 90 % of the arithmetic), 2 for the wide low-resolution passes.
template <int NT, int ROWS> __global__ void __launch_bounds__(256) k_conv3x3(const ConvParams p) {
    constexpr int TILE_H = 4 * ROWS, PATCH_H = TILE_H + 2;
    constexpr int WP = weight_pitch(NT);
    constexpr int W4 = 9 * CHUNK * WP / 4, W4_PER_THREAD = (W4 + 255) / 256;               // 16-byte pieces of a weight chunk
    constexpr int P4 = PATCH_H * PATCH_W * (CHUNK / 4), P4_PER_THREAD = (P4 + 255) / 256; // ... of a patch chunk
    __shared__ __attribute__((aligned(16))) float s_patch[PATCH_H * PATCH_W * PATCH_PITCH];
    __shared__ __attribute__((aligned(16))) float s_w[9 * CHUNK * WP];

    const int tiles_x = (p.w + TILE_W - 1) / TILE_W;
    const int tx0 = p.x0 + int(blockIdx.x % tiles_x) * TILE_W, ty0 = p.y0 + int(blockIdx.x / tiles_x) * TILE_H;
    const int tid = int(threadIdx.x), lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int r0 = ROWS * wave; // this wavefront's pixel rows of the tile

    f32x4 acc[ROWS][NT];
    for (int nt = 0; nt < NT; ++nt) {
        const float bias = p.bias[nt * 16 + m];
        for (int r = 0; r < ROWS; ++r) {
            acc[r][nt] = f32x4{bias, bias, bias, bias};
        }
    }

    const int chunks_a = p.a ? p.a_ch / CHUNK : 0, chunks_b = p.b ? p.b_ch / CHUNK : 0;
    const int n_chunks = chunks_a + chunks_b;

    // the patch pieces of this thread: piece e = tid + 256 j is channels 4 (e & 3) .. + 3 of patch pixel e >> 2.  Kept per piece: the
    // pixel's index in each input tensor (32 bits: a tensor has < 2^31 floats) and one bit "inside the tensor, border included"
    static_assert(P4_PER_THREAD <= 8, "piece indices are kept in 8-wide native vectors");
    i32x8 piece_a = {0, 0, 0, 0, 0, 0, 0, 0}, piece_b = piece_a; // (native vectors, indexed by unrolled constants: registers, not scratch)
    uint32_t inside = 0;
#pragma unroll
    for (int j = 0; j < P4_PER_THREAD; ++j) {
        const int e = tid + 256 * j, pixel = e >> 2;
        const int px = pixel % PATCH_W, py = pixel / PATCH_W;
        const int X = tx0 - 1 + px, Y = ty0 - 1 + py; // coordinates in this pass's resolution
        const bool in = e < P4 && X >= -1 && X <= p.in_w && Y >= -1 && Y <= p.in_h; // (the tensors carry a one-pixel zero border)
        inside |= in ? (1u << j) : 0u;
        const int sx = p.a_up ? (X >> 1) : X, sy = p.a_up ? (Y >> 1) : Y; // (arithmetic shift: -1 stays on the border)
        piece_a[j] = in ? sy * p.a_stride + sx : 0;
        piece_b[j] = in ? Y * p.b_stride + X : 0;
    }
    const int quad4 = (tid & 3) * 4; // (256 is a multiple of 4: every piece of a thread is the same channel quad)

    f32x4 held_patch[P4_PER_THREAD], held_w[W4_PER_THREAD]; // (native vectors: the arrays must become registers)
    // chunk `ch` -> registers
    auto fetch = [&](const int ch) __attribute__((always_inline)) {
        const bool from_a = ch < chunks_a;
        const float *src = from_a ? p.a + ch * CHUNK + quad4 : p.b + (ch - chunks_a) * CHUNK + quad4;
        const int n_ch = from_a ? p.a_ch : p.b_ch;
#pragma unroll
        for (int j = 0; j < P4_PER_THREAD; ++j) {
            held_patch[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if ((inside >> j) & 1u) {
                held_patch[j] = *reinterpret_cast<const f32x4 *>(src + ptrdiff_t(from_a ? piece_a[j] : piece_b[j]) * n_ch);
            }
        }
        const f32x4 *wsrc = reinterpret_cast<const f32x4 *>(p.weights + size_t(ch) * size_t(9 * CHUNK * WP));
#pragma unroll
        for (int j = 0; j < W4_PER_THREAD; ++j) {
            const int i = tid + 256 * j;
            held_w[j] = wsrc[(W4 % 256 == 0 || i < W4) ? i : W4 - 1]; // (the last round is partial: a clamped, unused load)
        }
    };
    // registers -> LDS
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < P4_PER_THREAD; ++j) {
            const int e = tid + 256 * j;
            if (P4 % 256 == 0 || e < P4) {
                *reinterpret_cast<f32x4 *>(&s_patch[(e >> 2) * PATCH_PITCH + quad4]) = held_patch[j];
            }
        }
#pragma unroll
        for (int j = 0; j < W4_PER_THREAD; ++j) {
            const int i = tid + 256 * j;
            if (W4 % 256 == 0 || i < W4) {
                reinterpret_cast<f32x4 *>(s_w)[i] = held_w[j];
            }
        }
    };

    fetch(0);
    for (int ch = 0; ch < n_chunks; ++ch) {
        __syncthreads(); // (the previous chunk's readers are done)
        commit();
        __syncthreads();
        if (ch + 1 < n_chunks) {
            fetch(ch + 1); // in flight while the matrix instructions below run
        }
        // 36 steps (tap, four channels) of ROWS x NT matrix instructions each.  The operands of step s + 1 are requested from LDS BEFORE the
        // matrix instructions of step s are issued and are waited for after them (the compiler, left alone, put every ds_read directly in
        // front of its first use: an LDS round trip in front of every group of matrix instructions); the scheduling barriers pin that order.
        float a_now[ROWS], b_now[NT], a_next[ROWS], b_next[NT];
        auto operands = [&](const int step, float (&a)[ROWS], float (&b)[NT]) __attribute__((always_inline)) {
            const int tap = step / (CHUNK / 4), k4 = step % (CHUNK / 4), ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                a[r] = s_patch[((r0 + r + ky) * PATCH_W + (m + kx)) * PATCH_PITCH + k4 * 4 + kq];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                b[nt] = s_w[(tap * CHUNK + k4 * 4 + kq) * WP + nt * 16 + m];
            }
        };
        operands(0, a_now, b_now);
#pragma unroll
        for (int step = 0; step < 9 * (CHUNK / 4); ++step) {
            if (step + 1 < 9 * (CHUNK / 4)) {
                operands(step + 1, a_next, b_next);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_now[r], b_now[nt], acc[r][nt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                a_now[r] = a_next[r];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                b_now[nt] = b_next[nt];
            }
        }
    }

    // ---- epilogue: lane holds out channel nt * 16 + m for the pixels x = tx0 + 4 kq + 0 .. 3 of rows ty0 + r0 .. + ROWS - 1
    const int xe = p.x0 + p.w, ye = p.y0 + p.h;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + m;
        if (n >= p.out_ch) {
            continue;
        }
        if (p.pool) {
#pragma unroll
            for (int rp = 0; rp < ROWS; rp += 2) { // (row pairs: r0 and ROWS are even)
                const int y = ty0 + r0 + rp;
                if (y >= ye) {
                    continue;
                }
                for (int pr = 0; pr < 2; ++pr) {
                    const int x = tx0 + 4 * kq + 2 * pr;
                    if (x < xe) {
                        float v = fmaxf(fmaxf(fmaxf(acc[rp][nt][2 * pr], 0.0f), fmaxf(acc[rp][nt][2 * pr + 1], 0.0f)),
                                        fmaxf(fmaxf(acc[rp + 1][nt][2 * pr], 0.0f), fmaxf(acc[rp + 1][nt][2 * pr + 1], 0.0f)));
                        v = fmaxf(v, 0.0f);
                        p.out[(ptrdiff_t(y / 2) * p.out_stride + (x / 2)) * p.out_ch + n] = v;
                    }
                }
            }
        } else {
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) {
                const int y = ty0 + r0 + rr;
                if (y >= ye) {
                    continue;
                }
                for (int q = 0; q < 4; ++q) {
                    const int x = tx0 + 4 * kq + q;
                    if (x >= xe) {
                        continue;
                    }
                    const float v = fmaxf(0.0f, acc[rr][nt][q]);
                    if (p.final_image) {
                        p.out[(ptrdiff_t(y) * p.out_stride + x) * 4 + n] = transfer_out_hdr(v);
                    } else {
                        p.out[(ptrdiff_t(y) * p.out_stride + x) * p.out_ch + n] = v;
                    }
                }
            }
        }
    }
}

// The renderer's three images as ONE 16-channel tensor (nine used: radiance through the HDR transfer function, base colour as it is,
// depth-normals as 0.5 n + 0.5; Convolution.h:222-244): what the first pass convolves and what dec_conv1a concatenates.  Rounds 1-3
// evaluated the transfer function (powf / logf) inside the convolution's staging, once per tile that touches the pixel and between the
// two barriers of the chunk; now it is evaluated once per pixel and the convolution stages the result like any other tensor.
// Pixels of the padded tensor outside the image stay zero (they are never written).
__global__ void __launch_bounds__(256) k_image_inputs(const float4 *__restrict__ full, const float4 *__restrict__ base, const float4 *__restrict__ dn,
                                                     const int w, const int h, float *__restrict__ out, const int out_stride) {
    const int n = w * h;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int x = i % w, y = i / w;
        const float4 a = full[i], b = base[i], c = dn[i];
        f32x4 *o = reinterpret_cast<f32x4 *>(out + (ptrdiff_t(y) * out_stride + x) * CHUNK);
        o[0] = f32x4{transfer_in_hdr(a.x), transfer_in_hdr(a.y), transfer_in_hdr(a.z), b.x};
        o[1] = f32x4{b.y, b.z, 0.5f * c.x + 0.5f, 0.5f * c.y + 0.5f};
        o[2] = f32x4{0.5f * c.z + 0.5f, 0.0f, 0.0f, 0.0f};
        o[3] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
}

hipError_t launch_image_inputs(const float4 *full, const float4 *base, const float4 *dn, const int w, const int h, float *out, const int out_stride,
                               const int blocks, hipStream_t stream) {
    k_image_inputs<<<blocks, 256, 0, stream>>>(full, base, dn, w, h, out, out_stride);
    return hipGetLastError();
}

hipError_t launch_conv(const ConvParams &p, const int n_tiles, hipStream_t stream) {
    if (n_tiles < 1 || n_tiles > 7 || p.w <= 0 || p.h <= 0) {
        return n_tiles < 1 || n_tiles > 7 ? hipErrorInvalidValue : hipSuccess;
    }
    const int rows = n_tiles <= 4 ? 4 : 2, tile_h = 4 * rows;
    const int tiles = ((p.w + TILE_W - 1) / TILE_W) * ((p.h + tile_h - 1) / tile_h);
    switch (n_tiles) {
    case 1:
        k_conv3x3<1, 4><<<tiles, 256, 0, stream>>>(p);
        break;
    case 2:
        k_conv3x3<2, 4><<<tiles, 256, 0, stream>>>(p);
        break;
    case 3:
        k_conv3x3<3, 4><<<tiles, 256, 0, stream>>>(p);
        break;
    case 4:
        k_conv3x3<4, 4><<<tiles, 256, 0, stream>>>(p);
        break;
    case 5:
        k_conv3x3<5, 2><<<tiles, 256, 0, stream>>>(p);
        break;
    case 6:
        k_conv3x3<6, 2><<<tiles, 256, 0, stream>>>(p);
        break;
    default:
        k_conv3x3<7, 2><<<tiles, 256, 0, stream>>>(p);
        break;
    }
    return hipGetLastError();
}


// ---- the f16 form (unet.h): the same tiling, v_mfma_f32_16x16x32_f16 ------------------------------------------------------------------------------
// Workgroup = 4 wavefronts = a (4 ROWS) x 16 tile of output pixels x all output channels, K walked in chunks of 32 input channels: per tap and
// chunk ONE matrix instruction per (pixel row, 16 output channels) where the f32 form issues four per 16 channels.
//   A operand  lane l: pixel m = l & 15 of the row, input channels 8 (l >> 4) .. + 7 of the chunk   (8 halves = one 16-byte LDS read)
//   B operand  lane l: out channel n = l & 15,      input channels 8 (l >> 4) .. + 7
//   D          as the f32 form: out channel l & 15, pixels 4 (l >> 4) + 0 .. 3
// (A and B distribute k over lanes and elements in the same way, so the sum over k does not depend on what that way is.)  LDS rows -- a patch
// pixel's 32 channels, an output channel's 32 weights of one tap -- are 64 bytes, no padding; their 16-byte units are swizzled (unet.h:
// swizzle_h) so that the operand reads (one ds_read_b128 each) are conflict-free.  The weights arrive from HBM already in that order (rayhip_unet_init), the patch is
// swizzled by its stagers.  Channel counts that are odd multiples of 16 (the 16-channel image tensor, 48, 80, 112) leave the upper half of their
// last chunk zero.  Staging is software-pipelined through registers like the f32 form's.
constexpr float H_MAX = 65504.0f; // largest finite half: what the f16 tensors saturate at
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

template <int NT, int ROWS> __global__ void __launch_bounds__(256, 2) k_conv3x3_h(const ConvParamsH p) {
    constexpr int TILE_H = 4 * ROWS, PATCH_H = TILE_H + 2;
    constexpr int P16 = PATCH_H * PATCH_W * 4, P16_PER_THREAD = (P16 + 255) / 256; // 16-byte pieces of a patch chunk (4 per pixel)
    constexpr int W16 = 9 * NT * 16 * 4, W16_PER_THREAD = (W16 + 255) / 256;       // ... of a weight chunk (4 per (tap, out channel))
    // LDS rows (a patch pixel's 32 channels; an out channel's 32 weights of a tap) are 64 bytes, their 16-byte units swizzled by the column
    // (unet.h: swizzle_h): a lane's operand is one conflict-free ds_read_b128 whose address is (a per-lane value that depends on kx only) +
    // (a compile-time offset for the row and the tap).
    constexpr int PATCH_LDS_W = PATCH_W;
    __shared__ __attribute__((aligned(16))) _Float16 s_patch[PATCH_H * PATCH_LDS_W * CHUNK_H];
    __shared__ __attribute__((aligned(16))) _Float16 s_w[9 * NT * 16 * CHUNK_H];

    const int tiles_x = (p.w + TILE_W - 1) / TILE_W;
    const int tx0 = p.x0 + int(blockIdx.x % tiles_x) * TILE_W, ty0 = p.y0 + int(blockIdx.x / tiles_x) * TILE_H;
    const int tid = int(threadIdx.x), lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int r0 = ROWS * wave;

    // (the weights are the matrix instruction's A operand, the pixels its B operand: D = W^T x patch^T, so a lane ends up with FOUR CONSECUTIVE out
    // channels 16 nt + 4 kq + 0 .. 3 of ONE pixel m of the row -- 8 contiguous bytes of the NHWC output, one store -- where the f32 form, pixels as A,
    // holds four pixels of one channel and stores them one 2- or 4-byte element at a time: first version of this kernel, 64 store instructions and
    // ~1000 instructions of address arithmetic per wavefront and tile, 40 % on top of the matrix phase)
    f32x4 acc[ROWS][NT];
    for (int nt = 0; nt < NT; ++nt) {
        const f32x4 bias = *reinterpret_cast<const f32x4 *>(p.bias + nt * 16 + 4 * kq);
        for (int r = 0; r < ROWS; ++r) {
            acc[r][nt] = bias;
        }
    }
    const int chunks_a = p.a ? (p.a_ch + CHUNK_H - 1) / CHUNK_H : 0, chunks_b = p.b ? (p.b_ch + CHUNK_H - 1) / CHUNK_H : 0;
    const int n_chunks = chunks_a + chunks_b;

    static_assert(P16_PER_THREAD <= 8, "piece indices are kept in 8-wide native vectors");
    i32x8 piece_a = {0, 0, 0, 0, 0, 0, 0, 0}, piece_b = piece_a;
    uint32_t inside = 0;
#pragma unroll
    for (int j = 0; j < P16_PER_THREAD; ++j) {
        const int e = tid + 256 * j, pixel = e >> 2;
        const int px = pixel % PATCH_W, py = pixel / PATCH_W;
        const int X = tx0 - 1 + px, Y = ty0 - 1 + py;
        const bool in = e < P16 && X >= -1 && X <= p.in_w && Y >= -1 && Y <= p.in_h;
        inside |= in ? (1u << j) : 0u;
        const int sx = p.a_up ? (X >> 1) : X, sy = p.a_up ? (Y >> 1) : Y;
        piece_a[j] = in ? sy * p.a_stride + sx : 0;
        piece_b[j] = in ? Y * p.b_stride + X : 0;
    }
    const int q = tid & 3; // (256 is a multiple of 4: every piece of a thread is the same quarter of its pixel's 32 channels)

    h8 held_patch[P16_PER_THREAD], held_w[W16_PER_THREAD];
    const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    bool held_have = false; // the chunk in the registers has channels for this thread's quarter (the upper half of a last, 16-channel chunk is zero)
    // chunk `ch` -> registers.  Every load is UNCONDITIONAL and from a valid address (a piece outside the tensor reads pixel 0, a quarter past the
    // tensor's channels the neighbouring pixel's: the allocations carry 64 bytes of slack); what must be zero is zeroed at commit time.  (A load
    // under `if (inside)` merges with the zero at the end of the branch, and the compiler waits for it THERE: six serialized round trips per
    // chunk and thread -- the wavefronts of the first version were parked in s_waitcnt 53 % of their time.)
    auto fetch = [&](const int ch) __attribute__((always_inline)) {
        const bool from_a = ch < chunks_a;
        const int n_ch = from_a ? p.a_ch : p.b_ch;
        const int c0 = (from_a ? ch : ch - chunks_a) * CHUNK_H + 8 * q; // first channel of this thread's pieces
        const _Float16 *src = static_cast<const _Float16 *>(from_a ? p.a : p.b) + c0;
        held_have = c0 < n_ch;
#pragma unroll
        for (int j = 0; j < P16_PER_THREAD; ++j) {
            held_patch[j] = *reinterpret_cast<const h8 *>(src + ptrdiff_t(from_a ? piece_a[j] : piece_b[j]) * n_ch);
        }
        const h8 *wsrc = reinterpret_cast<const h8 *>(static_cast<const _Float16 *>(p.weights) + size_t(ch) * size_t(9 * NT * 16 * CHUNK_H));
#pragma unroll
        for (int j = 0; j < W16_PER_THREAD; ++j) {
            const int i = tid + 256 * j;
            held_w[j] = wsrc[(W16 % 256 == 0 || i < W16) ? i : W16 - 1];
        }
    };
    // where this thread's patch pieces go in LDS (fixed over the chunks): unit q of the pixel in column `col` -> unit swizzle_h(col, q)
    i32x8 patch_dst = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < P16_PER_THREAD; ++j) {
        const int e = tid + 256 * j, pixel = e >> 2, col = pixel % PATCH_W, row = pixel / PATCH_W;
        patch_dst[j] = (row * PATCH_LDS_W + col) * CHUNK_H + swizzle_h(col, q) * 8;
    }
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < P16_PER_THREAD; ++j) {
            const int e = tid + 256 * j;
            if (P16 % 256 == 0 || e < P16) {
                *reinterpret_cast<h8 *>(&s_patch[patch_dst[j]]) = (held_have && ((inside >> j) & 1u)) ? held_patch[j] : zero8;
            }
        }
#pragma unroll
        for (int j = 0; j < W16_PER_THREAD; ++j) {
            const int i = tid + 256 * j;
            if (W16 % 256 == 0 || i < W16) {
                reinterpret_cast<h8 *>(s_w)[i] = held_w[j];
            }
        }
    };
    // a lane's operand = unit kq of a row: per-lane bases (A: one per kx, the wavefront's first row folded in; B: one), everything else of the
    // address is a compile-time offset
    const _Float16 *a_base[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int col = m + kx;
        a_base[kx] = s_patch + (r0 * PATCH_LDS_W + col) * CHUNK_H + swizzle_h(col, kq) * 8;
    }
    const _Float16 *b_base = s_w + m * CHUNK_H + swizzle_h(m, kq) * 8;
    fetch(0);
    for (int ch = 0; ch < n_chunks; ++ch) {
        __syncthreads();
        commit();
        __syncthreads();
        if (ch + 1 < n_chunks) {
            fetch(ch + 1);
        }
        // nine taps of ROWS x NT matrix instructions each.  The operands of tap t + 1 are requested from LDS BEFORE the matrix instructions of tap t
        // are issued and waited for after them (left alone, the compiler put every ds_read a few instructions in front of its use: an LDS round
        // trip per four or five matrix instructions); the scheduling barriers pin that order.
        h8 a_now[ROWS], b_now[NT], a_next[ROWS], b_next[NT];
        auto operands = [&](const int tap, h8 (&a)[ROWS], h8 (&b)[NT]) __attribute__((always_inline)) {
            constexpr int ROW_HALVES = PATCH_LDS_W * CHUNK_H;
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                a[r] = *reinterpret_cast<const h8 *>(a_base[kx] + (r + ky) * ROW_HALVES);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                b[nt] = *reinterpret_cast<const h8 *>(b_base + (tap * (NT * 16) + nt * 16) * CHUNK_H);
            }
        };
        operands(0, a_now, b_now);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) {
                operands(tap + 1, a_next, b_next);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    acc[r][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b_now[nt], a_now[r], acc[r][nt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                a_now[r] = a_next[r];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                b_now[nt] = b_next[nt];
            }
        }
    }

    // ---- epilogue: lane (m, kq) holds out channels 16 nt + 4 kq + 0 .. 3 of pixel x = tx0 + m in rows ty0 + r0 .. + ROWS - 1
    const int xe = p.x0 + p.w, ye = p.y0 + p.h;
    const int x = tx0 + m;
    if (p.final_image) { // 3 channels through the inverse HDR transfer into the float4 image (alpha stays): lanes kq = 0 of tile 0
        if (kq == 0 && x < xe) {
            float *out_f = static_cast<float *>(p.out);
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) {
                const int y = ty0 + r0 + rr;
                if (y < ye) {
                    float *o = out_f + (ptrdiff_t(y) * p.out_stride + x) * 4;
                    o[0] = transfer_out_hdr(fmaxf(0.0f, acc[rr][0][0])), o[1] = transfer_out_hdr(fmaxf(0.0f, acc[rr][0][1]));
                    o[2] = transfer_out_hdr(fmaxf(0.0f, acc[rr][0][2]));
                }
            }
        }
        return;
    }
    _Float16 *out_h = static_cast<_Float16 *>(p.out);
    if (p.pool) { // 2 x 2 max over rows (rp, rp + 1) -- this lane's registers -- and pixels (m, m ^ 1) -- the neighbouring lane; even m writes
#pragma unroll
        for (int rp = 0; rp < ROWS; rp += 2) {
            const int y = ty0 + r0 + rp;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                h4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float mine = fmaxf(fmaxf(acc[rp][nt][i], 0.0f), fmaxf(acc[rp + 1][nt][i], 0.0f));
                    v[i] = _Float16(fminf(fmaxf(mine, __shfl_xor(mine, 1)), H_MAX)); // (saturating: an activation beyond the half range stays finite)
                }
                if ((m & 1) == 0 && x < xe && y < ye && nt * 16 < p.out_ch) {
                    *reinterpret_cast<h4 *>(out_h + (ptrdiff_t(y / 2) * p.out_stride + (x / 2)) * p.out_ch + nt * 16 + 4 * kq) = v;
                }
            }
        }
    } else if (x < xe) {
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int y = ty0 + r0 + rr;
            if (y >= ye) {
                continue;
            }
            _Float16 *o = out_h + (ptrdiff_t(y) * p.out_stride + x) * p.out_ch + 4 * kq;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (nt * 16 < p.out_ch) {
                    // ReLU and saturation at the largest half in one clamp: an activation beyond 65504 stays finite
                    *reinterpret_cast<h4 *>(o + nt * 16) = h4{_Float16(fminf(fmaxf(0.0f, acc[rr][nt][0]), H_MAX)), _Float16(fminf(fmaxf(0.0f, acc[rr][nt][1]), H_MAX)),
                                                             _Float16(fminf(fmaxf(0.0f, acc[rr][nt][2]), H_MAX)), _Float16(fminf(fmaxf(0.0f, acc[rr][nt][3]), H_MAX))};
                }
            }
        }
    }
}

// the renderer's three images as one 16-channel f16 tensor (k_image_inputs above, halves)
__global__ void __launch_bounds__(256) k_image_inputs_h(const float4 *__restrict__ full, const float4 *__restrict__ base, const float4 *__restrict__ dn,
                                                       const int w, const int h, _Float16 *__restrict__ out, const int out_stride) {
    const int n = w * h;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int x = i % w, y = i / w;
        const float4 a = full[i], b = base[i], c = dn[i];
        h8 *o = reinterpret_cast<h8 *>(out + (ptrdiff_t(y) * out_stride + x) * CHUNK);
        o[0] = h8{_Float16(transfer_in_hdr(a.x)), _Float16(transfer_in_hdr(a.y)), _Float16(transfer_in_hdr(a.z)), _Float16(b.x),
                  _Float16(b.y), _Float16(b.z), _Float16(0.5f * c.x + 0.5f), _Float16(0.5f * c.y + 0.5f)};
        o[1] = h8{_Float16(0.5f * c.z + 0.5f), 0, 0, 0, 0, 0, 0, 0};
    }
}

hipError_t launch_image_inputs_h(const float4 *full, const float4 *base, const float4 *dn, const int w, const int h, void *out, const int out_stride,
                                 const int blocks, hipStream_t stream) {
    k_image_inputs_h<<<blocks, 256, 0, stream>>>(full, base, dn, w, h, static_cast<_Float16 *>(out), out_stride);
    return hipGetLastError();
}

hipError_t launch_conv_h(const ConvParamsH &p, const int n_tiles, hipStream_t stream) {
    if (n_tiles < 1 || n_tiles > 7 || p.w <= 0 || p.h <= 0) {
        return n_tiles < 1 || n_tiles > 7 ? hipErrorInvalidValue : hipSuccess;
    }
    const int rows = n_tiles <= 4 ? 4 : 2, tile_h = 4 * rows;
    const int tiles = ((p.w + TILE_W - 1) / TILE_W) * ((p.h + tile_h - 1) / tile_h);
    switch (n_tiles) {
    case 1:
        k_conv3x3_h<1, 4><<<tiles, 256, 0, stream>>>(p);
        break;
    case 2:
        k_conv3x3_h<2, 4><<<tiles, 256, 0, stream>>>(p);
        break;
    case 3:
        k_conv3x3_h<3, 4><<<tiles, 256, 0, stream>>>(p);
        break;
    case 4:
        k_conv3x3_h<4, 4><<<tiles, 256, 0, stream>>>(p);
        break;
    case 5:
        k_conv3x3_h<5, 2><<<tiles, 256, 0, stream>>>(p);
        break;
    case 6:
        k_conv3x3_h<6, 2><<<tiles, 256, 0, stream>>>(p);
        break;
    default:
        k_conv3x3_h<7, 2><<<tiles, 256, 0, stream>>>(p);
        break;
    }
    return hipGetLastError();
}

} // namespace unet
} // namespace rt
