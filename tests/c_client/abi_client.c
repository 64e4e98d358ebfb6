/* A caller of libpiscoseg.so written in plain C: no Python, no torch -- the drop-in boundary as a C program sees it.
 * Builds with gcc against include/piscoseg.h and the HIP runtime's C API (device memory only).
 *
 *   abi_client <H> <W> <n> <seed>
 *
 * renders n synthetic frames on the device (pseg_synth_frames with the object list read from stdin: "nobj" then nobj
 * lines of 8 integers, then n + 1 frame offsets), runs the fused path with a disk mask and prints one line per result
 * that tests/test_gpu_c_client.py compares with what the Python binding gives for the same input:
 *   rows <n_rows> saved <n_saved> crop_bytes <bytes>
 *   frame <f> mean <%.17g> std <%.17g> valid <0|1> rows <k>
 *   row <i> frame c label area x y w h saved
 *   props <s> <32 values %.17g>
 *   cleaned_sum <sum of all cleaned bytes> labels_sum <sum of all label values>
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "../../include/piscoseg.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, ctx ? pseg_last_error(ctx) : ""); return 2; } } while (0)
#define HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 3; } } while (0)

int main(int argc, char **argv)
{
    if (argc != 5) { fprintf(stderr, "usage: abi_client H W n seed\n"); return 1; }
    const int H = atoi(argv[1]), W = atoi(argv[2]), n = atoi(argv[3]);
    const uint64_t seed = (uint64_t)strtoull(argv[4], NULL, 10);
    pseg_ctx *ctx = NULL;
    int nobj = 0;
    if (scanf("%d", &nobj) != 1 || nobj < 0) return 1;
    int32_t *objs = (int32_t *)malloc((size_t)(nobj > 0 ? nobj : 1) * 8 * sizeof(int32_t));
    int32_t *off = (int32_t *)malloc((size_t)(n + 1) * sizeof(int32_t));
    for (int i = 0; i < nobj * 8; i++) if (scanf("%d", &objs[i]) != 1) return 1;
    for (int i = 0; i <= n; i++) if (scanf("%d", &off[i]) != 1) return 1;

    const size_t P = (size_t)H * W;
    uint8_t *frames = NULL, *cleaned = NULL, *mask = NULL;
    int32_t *labels = NULL;
    HIP(hipSetDevice(0));
    CHECK(pseg_create(0, H, W, n, &ctx));
    HIP(hipMalloc((void **)&frames, n * P));
    HIP(hipMalloc((void **)&cleaned, n * P));
    HIP(hipMalloc((void **)&mask, P));
    HIP(hipMalloc((void **)&labels, n * P * sizeof(int32_t)));
    CHECK(pseg_synth_frames(ctx, n, 0, seed, objs, off, frames));
    CHECK(pseg_disk_mask(ctx, (H < W ? H : W) / 2 - 6, mask));

    double *mean_std = (double *)malloc((size_t)n * 2 * sizeof(double));
    uint8_t *valid = (uint8_t *)malloc((size_t)n);
    pseg_table t;
    CHECK(pseg_process(ctx, frames, n, 6, mask, 10, 10.0, 150.0, cleaned, mean_std, valid, labels, &t));

    printf("rows %d saved %d crop_bytes %lld\n", t.n_rows, t.n_saved, (long long)t.crop_bytes);
    for (int f = 0; f < n; f++)
        printf("frame %d mean %.17g std %.17g valid %d rows %d\n", f, mean_std[2 * f], mean_std[2 * f + 1], (int)valid[f],
               t.frame_row_offset[f + 1] - t.frame_row_offset[f]);
    for (int i = 0; i < t.n_rows; i++)
        printf("row %d %d %d %d %.17g %d %d %d %d %d\n", i, t.row_frame[i], t.row_c[i], t.row_label[i], t.row_area[i], t.row_x[i], t.row_y[i],
               t.row_w[i], t.row_h[i], (int)t.row_saved[i]);
    for (int s = 0; s < t.n_saved; s++) {
        printf("props %d", s);
        for (int k = 0; k < PSEG_NPROPS; k++) printf(" %.17g", t.saved_props[(size_t)s * PSEG_NPROPS + k]);
        printf("\n");
    }
    {   /* the device outputs, summed on the host */
        uint8_t *hc = (uint8_t *)malloc(n * P);
        int32_t *hl = (int32_t *)malloc(n * P * sizeof(int32_t));
        unsigned long long cs = 0, ls = 0;
        HIP(hipMemcpy(hc, cleaned, n * P, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(hl, labels, n * P * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int f = 0; f < n; f++) {
            if (!valid[f]) continue;                        /* contents of corrupt frames are unspecified */
            for (size_t i = 0; i < P; i++) { cs += hc[(size_t)f * P + i]; ls += (unsigned long long)hl[(size_t)f * P + i]; }
        }
        printf("cleaned_sum %llu labels_sum %llu\n", cs, ls);
        free(hc); free(hl);
    }
    /* error path: a batch longer than the context allows comes back as a code and a message, nothing aborts */
    {
        int rc = pseg_process(ctx, frames, n + 1, 6, mask, 10, 10.0, 150.0, cleaned, mean_std, valid, labels, &t);
        printf("too_long rc %d text %s\n", rc, pseg_last_error(ctx));
    }
    CHECK(pseg_destroy(ctx));
    hipFree(frames); hipFree(cleaned); hipFree(mask); hipFree(labels);
    free(objs); free(off); free(mean_std); free(valid);
    return 0;
}
