/*
 * ref_shim.h -- oracle-only C entry points into the REAL reference (libray_ref.so, see oracle/Makefile).
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; nothing under ray_amd/ does.  The library also exports the generic ray_* C view of the public API
 * (ray_amd/host/ray_capi.h) restricted to the reference's CPU backends.
 *
 * The refk_* functions call the reference's own kernel-level functions (file:line cited per function) so that
 * the HIP kernels can be checked stage by stage, not only through final images.
 */
#ifndef REF_SHIM_H
#define REF_SHIM_H

#include <stddef.h>
#include <stdint.h>

#include "../include/rayhip.h"
#include "../ray_amd/host/ray_capi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the PMJ02 table the reference renders with (internal/precomputed/__pmj02_samples.inl, Core.h:363-368) */
void refk_pmj_table(const uint32_t **out_ptr, uint32_t *out_count);

/* Cpu::Renderer::UpdateFilterTable (RendererCPU.h:1234-1258): 1024-entry inverse CDF */
void refk_filter_table(uint32_t pixel_filter, float filter_width, float *out_table /*[1024]*/);

/* Flat arrays of a finalized scene + its current camera + filter table, serialised with
 * ray_amd/csrc/scene_blob.h.  The scene must have been created by a "REF" renderer (2-wide BVH).
 * Returns a malloc'd buffer (free with refk_free). */
int refk_export_scene(ray_scene *s, void **out_blob, size_t *out_size);
void refk_free(void *p);

/* Ref::GeneratePrimaryRays, CoreRef.cpp:1429-1553 (required_samples = NULL) */
int refk_generate_primary_rays(ray_scene *s, int w, int h, const int rect[4], int iteration, rayhip_ray *out_rays,
                               rayhip_hit *out_hits, int *out_count);
/* Ref::IntersectScene (closest), CoreRef.cpp:3041-3158; rays/hits in-out */
int refk_intersect_closest(ray_scene *s, rayhip_ray *rays, rayhip_hit *hits, int count, int iteration);
/* Ref::IntersectScene(shadow_ray_t), CoreRef.cpp:3160-3262 */
int refk_intersect_shadow(ray_scene *s, const rayhip_shadow_ray *rays, int count, int iteration, float *out_rc);
/* Ref::get_scrambled_2d_rand, CoreRef.cpp:1418-1427 */
void refk_scrambled_rand(const uint32_t *dims, const uint32_t *seeds, const int32_t *samples, int count, float *out_xy);
/* Ref::ShadePrimary (bounce == 0) / Ref::ShadeSecondary (bounce >= 1) on caller-provided rays+hits
 * (ShadeRef.cpp:1654-1738).  out_color: w*h*4 in-out (temp buffer); secondary / shadow rays are returned. */
int refk_shade(ray_scene *s, int w, int h, int bounce, int iteration, const rayhip_ray *rays, const rayhip_hit *hits,
               int count, float *inout_color, rayhip_ray *out_secondary, int *out_secondary_count,
               rayhip_shadow_ray *out_shadow, int *out_shadow_count);

/* The UNet denoiser pass by pass, on caller-provided images: the schedule of Cpu::Renderer::DenoiseImage(pass, region)
 * (RendererCPU.h:790-1007) with the reference's own Ref::Convolution3x3 / ConvolutionConcat3x3 / ClearBorders
 * (DenoiseRef.cpp, Convolution.h), weights from SetupUNetWeights<float>(8) and tensor sizes from SetupUNetFilter (no
 * aliasing), full-frame region.  Runs passes 0 .. last_pass and copies out the tensor pass `last_pass` wrote -- including
 * its one-pixel border, NHWC, dims = {rows, columns, channels} -- or, for pass 15, the w*h*4 filtered image (alpha taken
 * from `full`).  Returns the number of floats written, 0 on error.  (The end-to-end path through the renderer is
 * ray_renderer_init_unet / ray_renderer_denoise_unet; tests check that the two agree bit for bit.) */
/* SetupUNetWeights<float>(8, &offsets, weights) (UNetFilter.cpp:296-570): the weight blob Cpu::Renderer::InitUNetFilter keeps and
 * unet_weight_offsets_t as 32 ints; returns the number of floats (call with out == NULL for the size) */
int refk_unet_weights(float *out, int capacity, int32_t out_offsets[32]);
size_t refk_unet_passes(int w, int h, const float *full_rgba, const float *base_rgba, const float *depth_normals_rgba, int last_pass,
                        float *out, size_t capacity, int out_dims[3]);

#ifdef __cplusplus
}
#endif

#endif
