// unet.h -- the UNet denoiser's convolution pass on MFMA: interface between the launcher (rayhip.hip: rayhip_unet_init,
// rayhip_denoise_unet) and the kernels (unet_kernels.hip, its own translation unit).
//
// Reference: Cpu::Renderer::DenoiseImage(int pass, const RegionContext &) -- sixteen passes, RendererCPU.h:790-1007 -- over
// Convolution3x3 / ConvolutionConcat3x3 (Convolution.h:117-589; GPU twin shaders/convolution.comp.glsl, whose
// cooperative-matrix variant is the one place the reference itself uses matrix hardware).  The network is OIDN's UNet:
// 3 x 3 convolutions with bias + ReLU over NHWC fp32 tensors that carry a one-pixel zero border, 2 x 2 max-pooling on the way
// down, nearest-neighbour upsampling + concatenation with the matching encoder tensor on the way up, an HDR transfer
// function on the radiance input and its inverse on the output.
//
// Every pass is ONE implicit GEMM: D[pixel][out channel] = sum over k = (tap, in channel) of A[pixel][k] * W[k][out channel],
// executed with v_mfma_f32_16x16x4_f32 -- f32 in, f32 accumulate: the exact-f32 matrix instruction (157 TFLOP/s, the vector
// rate), chosen over the f16 forms because parity is the first gate: against the reference's fp32 arithmetic only the
// ORDER of the additions differs (tests/test_gpu_unet.py: every pass within 2e-5 of the oracle's tensor).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {
namespace unet {

constexpr int TILE_W = 16;              // output pixels per row of a workgroup's tile (its height: 4 wavefronts x 4 or 2 rows, unet_kernels.hip)
constexpr int CHUNK = 16;              // input channels staged through LDS at a time

// padded out-channel pitch of the weight rows in LDS and HBM: a multiple of 16 whose residue mod 64 is 16 or 48, so that the
// four k-rows a wave-wide B fetch touches fall into four different 16-bank groups
__host__ __device__ constexpr int weight_pitch(const int n_tiles) { return ((n_tiles * 16) % 64 == 16 || (n_tiles * 16) % 64 == 48) ? n_tiles * 16 : n_tiles * 16 + 16; }

struct ConvParams {
    // first input: a tensor (interior pointer: pixel (0, 0) behind the border), optionally read through a nearest-neighbour upsample
    const float *a;
    int a_stride, a_ch, a_up;
    // second input (concatenated behind the first): another tensor ...
    const float *b;
    int b_stride, b_ch;
    // (the renderer's three images enter as a tensor too: launch_image_inputs writes them as 16 channels, nine used)
    const float *weights; // [chunk][tap][CHUNK][weight_pitch] (rayhip.hip: repack_conv)
    const float *bias;    // [weight_pitch]
    float *out;           // interior pointer of the output tensor, or the float4 image of the last pass
    int out_stride, out_ch;
    int x0, y0, w, h; // the rect of this pass in its own resolution
    int in_w, in_h;   // interior size of the inputs at this pass's resolution (reads outside [-1, in_w] x [-1, in_h] are zero)
    int pool;         // 2 x 2 max pooling: one output pixel per 2 x 2 block, at (y / 2, x / 2)
    int final_image;  // last pass: 3 channels through the inverse HDR transfer into a float4 image of pitch out_stride
};

// the three images (radiance through the HDR transfer, base colour, depth-normals as 0.5 n + 0.5) -> the interior of a 16-channel tensor
hipError_t launch_image_inputs(const float4 *full, const float4 *base, const float4 *dn, int w, int h, float *out, int out_stride, int blocks,
                               hipStream_t stream);

// n_tiles = out channels / 16 rounded up (1 .. 7)
hipError_t launch_conv(const ConvParams &p, int n_tiles, hipStream_t stream);

} // namespace unet
} // namespace rt
