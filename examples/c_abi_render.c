/* c_abi_render.c -- the C ABI of librayhip used from plain C: what a host in any language does through its FFI.
 *
 *   cc -std=c99 -I include examples/c_abi_render.c -L ray_amd/csrc/_build -lrayhip -Wl,-rpath,$PWD/ray_amd/csrc/_build -lm -o c_abi_render
 *   ./c_abi_render tests/golden/cornell_basic.rayscene tests/golden/pmj02_samples.npy 256 256 64 out.ppm
 *
 * A scene travels as a blob (ray_amd/csrc/scene_blob.h; written by SceneHIP or by tests/golden/make_fixtures.py): flat arrays
 * in the reference's layouts + camera + filter table.  The PMJ02 sample table is the reference's (Core.h:363-368); here it
 * is read from the .npy the tests use (a 128-byte header, then 32 x 4096 x 2 uint32).
 * Reference call sequence this mirrors: samples/00_basic/main.cpp (CreateRenderer, CreateScene ... RenderScene x N,
 * get_pixels_ref). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rayhip.h"

static void *read_file(const char *path, size_t *size, size_t align) {
    FILE *f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "cannot open %s\n", path);
        return NULL;
    }
    fseek(f, 0, SEEK_END);
    *size = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    void *p = NULL;
    if (posix_memalign(&p, align, *size ? *size : align) != 0 || fread(p, 1, *size, f) != *size) {
        fprintf(stderr, "cannot read %s\n", path);
        fclose(f);
        free(p);
        return NULL;
    }
    fclose(f);
    return p;
}

#define TRY(call)                                                                                                      \
    if ((call) != 0) {                                                                                                 \
        fprintf(stderr, "%s failed: %s\n", #call, rayhip_last_error());                                                \
        return 1;                                                                                                      \
    }

int main(int argc, char **argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: %s scene.rayscene pmj02_samples.npy width height spp out.ppm\n", argv[0]);
        return 2;
    }
    const int w = atoi(argv[3]), h = atoi(argv[4]), spp = atoi(argv[5]);
    if (rayhip_device_count() < 1) {
        fprintf(stderr, "no HIP device: librayhip has no CPU path\n");
        return 3;
    }
    size_t blob_size = 0, npy_size = 0;
    void *blob = read_file(argv[1], &blob_size, 64); /* (blobs must be 16-byte aligned) */
    uint8_t *npy = (uint8_t *)read_file(argv[2], &npy_size, 64);
    if (!blob || !npy) {
        return 1;
    }
    /* .npy v1: magic (6) + version (2) + header length (2, little endian) + header; data follows */
    const size_t data_off = 10u + (size_t)npy[8] + ((size_t)npy[9] << 8);
    const uint32_t pmj_count = (uint32_t)((npy_size - data_off) / 4u);

    rayhip_ctx *ctx = NULL;
    rayhip_camera cam;
    TRY(rayhip_ctx_create(0, &ctx))
    char name[128];
    TRY(rayhip_ctx_device_name(ctx, name, (int)sizeof name))
    TRY(rayhip_upload_static(ctx, (const uint32_t *)(npy + data_off), pmj_count))
    TRY(rayhip_resize(ctx, w, h))
    TRY(rayhip_scene_upload_blob(ctx, blob, blob_size, &cam)) /* uploads the arrays and the filter table, returns the camera */
    const int rect[4] = {0, 0, w, h};
    /* iterations 1..spp in as few wavefront passes as the library sees fit (bit-identical to spp x rayhip_render) */
    TRY(rayhip_render_batch(ctx, &cam, rect, 1, spp, 0, NULL))
    float *rgba = (float *)malloc((size_t)w * (size_t)h * 16u); /* allocate read-back buffers once and keep them */
    TRY(rayhip_readback(ctx, RAYHIP_BUF_FINAL, rgba, w))       /* tone-mapped; RAYHIP_BUF_RAW is the linear running mean */
    TRY(rayhip_sync(ctx))

    FILE *out = fopen(argv[6], "wb");
    if (!out) {
        fprintf(stderr, "cannot write %s\n", argv[6]);
        return 1;
    }
    fprintf(out, "P6\n%d %d\n255\n", w, h);
    double mean = 0.0;
    for (int i = 0; i < w * h; ++i) {
        for (int c = 0; c < 3; ++c) {
            float v = rgba[4 * i + c];
            v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
            mean += v;
            fputc((int)(v * 255.0f + 0.5f), out);
        }
    }
    fclose(out);
    printf("%s: %dx%d, %d spp, mean of the tone-mapped frame %.6f -> %s\n", name, w, h, spp, mean / (3.0 * w * h), argv[6]);
    rayhip_ctx_destroy(ctx);
    free(rgba);
    free(blob);
    free(npy);
    return 0;
}
