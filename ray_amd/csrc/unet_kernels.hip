// unet_kernels.hip -- one 3 x 3 convolution pass of the UNet denoiser as an implicit GEMM on the f32 matrix cores (unet.h).
//
// Workgroup = 4 wavefronts = an 8 x 16 tile of output pixels x ALL output channels.  The K dimension (9 taps x input
// channels) is walked in chunks of 16 channels: the chunk's (8 + 2) x (16 + 2) input patch and its 9 x 16 x N weight rows
// are staged in LDS (patch pixels 17 floats apart, weight rows at a pitch that spreads the four k-rows of a fetch over the
// four 16-bank groups: both operand fetches of a wavefront are conflict-free), then every wavefront issues, per tap and per
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

constexpr int PATCH_W = TILE_W + 2, PATCH_H = TILE_H + 2;
constexpr int PATCH_PITCH = CHUNK + 1; // floats per staged pixel (odd: 16 pixels x 4 channels land in distinct banks)

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

template <int NT> __global__ void __launch_bounds__(256) k_conv3x3(const ConvParams p) {
    constexpr int WP = weight_pitch(NT);
    __shared__ float s_patch[PATCH_H * PATCH_W * PATCH_PITCH];
    __shared__ float s_w[9 * CHUNK * WP];

    const int tiles_x = (p.w + TILE_W - 1) / TILE_W;
    const int tx0 = p.x0 + int(blockIdx.x % tiles_x) * TILE_W, ty0 = p.y0 + int(blockIdx.x / tiles_x) * TILE_H;
    const int lane = int(threadIdx.x) & 63, wave = int(threadIdx.x) >> 6;
    const int m = lane & 15, kq = lane >> 4;
    const int r0 = 2 * wave; // this wavefront's two pixel rows of the tile

    f32x4 acc[2][NT];
    for (int nt = 0; nt < NT; ++nt) {
        const float bias = p.bias[nt * 16 + m];
        acc[0][nt] = f32x4{bias, bias, bias, bias};
        acc[1][nt] = acc[0][nt];
    }

    const int chunks_a = p.a ? p.a_ch / CHUNK : 0, chunks_b = p.b ? p.b_ch / CHUNK : 0, chunks_img = p.img_full ? 1 : 0;
    const int n_chunks = chunks_a + chunks_b + chunks_img;
    for (int ch = 0; ch < n_chunks; ++ch) {
        __syncthreads(); // (the previous chunk's readers are done)
        // ---- stage the input patch of this chunk: PATCH_H x PATCH_W pixels x CHUNK channels
        for (int i = int(threadIdx.x); i < PATCH_H * PATCH_W * CHUNK; i += 256) {
            const int c = i % CHUNK, px = (i / CHUNK) % PATCH_W, py = i / (CHUNK * PATCH_W);
            const int X = tx0 - 1 + px, Y = ty0 - 1 + py; // coordinates in this pass's resolution
            float v = 0.0f;
            if (ch < chunks_a) {
                if (X >= -1 && X <= p.in_w && Y >= -1 && Y <= p.in_h) {
                    const int sx = p.a_up ? (X >> 1) : X, sy = p.a_up ? (Y >> 1) : Y; // (arithmetic shift: -1 stays on the border)
                    v = p.a[(ptrdiff_t(sy) * p.a_stride + sx) * p.a_ch + ch * CHUNK + c];
                }
            } else if (ch < chunks_a + chunks_b) {
                if (X >= -1 && X <= p.in_w && Y >= -1 && Y <= p.in_h) {
                    v = p.b[(ptrdiff_t(Y) * p.b_stride + X) * p.b_ch + (ch - chunks_a) * CHUNK + c];
                }
            } else if (c < 9 && X >= 0 && X < p.img_w && Y >= 0 && Y < p.img_h) {
                const size_t idx = size_t(Y) * size_t(p.img_w) + size_t(X);
                const float4 *img = c < 3 ? p.img_full : (c < 6 ? p.img_base : p.img_dn);
                const float4 t = img[idx];
                const int cc = c % 3;
                const float s = cc == 0 ? t.x : (cc == 1 ? t.y : t.z);
                v = c < 3 ? transfer_in_hdr(s) : (c < 6 ? s : 0.5f * s + 0.5f);
            }
            s_patch[(py * PATCH_W + px) * PATCH_PITCH + c] = v;
        }
        // ---- and its weights: 9 taps x CHUNK channels x WP, contiguous in HBM
        {
            const float *src = p.weights + size_t(ch) * size_t(9 * CHUNK * WP);
            for (int i = int(threadIdx.x); i < 9 * CHUNK * WP; i += 256) {
                s_w[i] = src[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int k4 = 0; k4 < CHUNK / 4; ++k4) {
                const float a0 = s_patch[((r0 + ky) * PATCH_W + (m + kx)) * PATCH_PITCH + k4 * 4 + kq];
                const float a1 = s_patch[((r0 + 1 + ky) * PATCH_W + (m + kx)) * PATCH_PITCH + k4 * 4 + kq];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float b = s_w[(tap * CHUNK + k4 * 4 + kq) * WP + nt * 16 + m];
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[1][nt], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: lane holds out channel nt * 16 + m for the pixels x = tx0 + 4 kq + 0 .. 3 of rows ty0 + r0, ty0 + r0 + 1
    const int xe = p.x0 + p.w, ye = p.y0 + p.h;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + m;
        if (n >= p.out_ch) {
            continue;
        }
        if (p.pool) {
            const int y = ty0 + r0;
            if (y < ye) {
                for (int pr = 0; pr < 2; ++pr) {
                    const int x = tx0 + 4 * kq + 2 * pr;
                    if (x < xe) {
                        float v = fmaxf(fmaxf(fmaxf(acc[0][nt][2 * pr], 0.0f), fmaxf(acc[0][nt][2 * pr + 1], 0.0f)),
                                        fmaxf(fmaxf(acc[1][nt][2 * pr], 0.0f), fmaxf(acc[1][nt][2 * pr + 1], 0.0f)));
                        v = fmaxf(v, 0.0f);
                        p.out[(ptrdiff_t(y / 2) * p.out_stride + (x / 2)) * p.out_ch + n] = v;
                    }
                }
            }
        } else {
            for (int rr = 0; rr < 2; ++rr) {
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

hipError_t launch_conv(const ConvParams &p, const int n_tiles, hipStream_t stream) {
    const int tiles = ((p.w + TILE_W - 1) / TILE_W) * ((p.h + TILE_H - 1) / TILE_H);
    if (tiles <= 0) {
        return hipSuccess;
    }
    switch (n_tiles) {
    case 1:
        k_conv3x3<1><<<tiles, 256, 0, stream>>>(p);
        break;
    case 2:
        k_conv3x3<2><<<tiles, 256, 0, stream>>>(p);
        break;
    case 3:
        k_conv3x3<3><<<tiles, 256, 0, stream>>>(p);
        break;
    case 4:
        k_conv3x3<4><<<tiles, 256, 0, stream>>>(p);
        break;
    case 5:
        k_conv3x3<5><<<tiles, 256, 0, stream>>>(p);
        break;
    case 6:
        k_conv3x3<6><<<tiles, 256, 0, stream>>>(p);
        break;
    case 7:
        k_conv3x3<7><<<tiles, 256, 0, stream>>>(p);
        break;
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace unet
} // namespace rt
