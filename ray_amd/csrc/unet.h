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

// ---- the f16 form (round 5): f16 tensors and weights, f32 accumulators, v_mfma_f32_16x16x32_f16 ------------------------------------------
// What the reference's own GPU backends run where the device has half-precision matrix hardware (internal/RendererVK.cpp:254-263, 1834-1844:
// the convolution_*_fp16 / *_coop_16x16x16_CF16 shader sets; tensors and weights are 16-bit there too, RendererGPU.h:533-545, 569, 632).  The f32
// form above stays the exact one (parity against the CPU oracle to 2e-5); this one is 12.8 x the matrix rate and half the tensor traffic, and is
// held to an fp16-appropriate bound against the same oracle (tests/test_gpu_unet.py).
constexpr int CHUNK_H = 32; // input channels staged through LDS at a time: one v_mfma_f32_16x16x32_f16 per (tap, chunk, 16 x 16 output tile)

struct ConvParamsH {
    const void *a; // f16 tensors (interior pointers), channel counts multiples of 16
    int a_stride, a_ch, a_up;
    const void *b;
    int b_stride, b_ch;
    const void *weights; // f16 [chunk of 32 in][tap][out channel (16 n_tiles)][32 in], the 16-byte units of a row swizzled as the kernel reads them (swizzle_h)
    const float *bias;   // [16 n_tiles] f32
    void *out;           // f16 tensor interior, or the float4 image of the last pass
    int out_stride, out_ch;
    int x0, y0, w, h, in_w, in_h, pool, final_image;
};
// position (in 16-byte units, 0 .. 3) of unit `u` (input channels 8 u .. 8 u + 7 of a 32-channel row) of the row in column `col` (a patch pixel's
// column, or an out channel for the weights) in LDS.  Rows are 64 bytes, no padding; a lane's A or B operand is ONE ds_read_b128.  That
// instruction serves its 64 lanes in four groups of 16 -- lanes {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32
// (MI355X_MICROARCH.md, LDS table) -- i.e. columns c, c + 12 with operand quarter kq and columns c + 4, c + 8 with quarter kq ^ 1 share four
// 16-byte bank slots; u ^ 2 ((col >> 2) & 1) puts the four on different ones whatever c is (the only functions of (col >> 2) & 3 that do, by
// enumeration), so the A reads of all three horizontal taps and the B reads are free of bank conflicts.  (First version: 8-byte pieces swizzled
// for ds_read_b64's 2 x 32-lane service -- the compiler fused the pairs into ds_read2_b64, which is served 4 x 16 lanes on 32 banks: 44 % of
// the LDS cycles were conflicts, profiles/r05/unet_f16_pmc_first_version.txt.)
__host__ __device__ constexpr int swizzle_h(const int col, const int u) { return u ^ (((col >> 2) & 1) << 1); }

hipError_t launch_image_inputs_h(const float4 *full, const float4 *base, const float4 *dn, int w, int h, void *out, int out_stride, int blocks,
                                 hipStream_t stream);
hipError_t launch_conv_h(const ConvParamsH &p, int n_tiles, hipStream_t stream);

} // namespace unet
} // namespace rt
