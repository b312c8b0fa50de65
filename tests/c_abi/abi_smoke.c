/* A plain C program against the C ABI of libsubpixal_hip.so: no Python, no torch.  It is what a
 * binding in any language does: device memory from the HIP runtime, plain pointers and sizes into
 * the spx_* entry points, results copied back.  Built and run by tests/test_gpu_c_abi.py:
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ abi_smoke.c -I include -I /opt/rocm/include -L subpixal_amd/csrc
 *       -L /opt/rocm/lib -lsubpixal_hip -lamdhip64 -lm
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "subpixal_hip.h"

#define CHECK_HIP(call)                                                          \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));           \
            return 2;                                                            \
        }                                                                        \
    } while (0)
#define CHECK_SPX(call)                                                          \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != 0) {                                                          \
            fprintf(stderr, "%s: %d %s\n", #call, rc_, spx_last_error());        \
            return 3;                                                            \
        }                                                                        \
    } while (0)

static int run_pairs(int n, int upsample, int64_t count, double tol) {
    const size_t npx = (size_t)n * n;
    float *ref, *img;
    double *truth, *out;
    int32_t* status;
    void* ws = NULL;
    CHECK_HIP(hipMalloc((void**)&ref, count * npx * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&img, count * npx * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&truth, count * 2 * sizeof(double)));
    CHECK_HIP(hipMalloc((void**)&out, count * 2 * sizeof(double)));
    CHECK_HIP(hipMalloc((void**)&status, count * sizeof(int32_t)));
    const size_t ws_bytes = spx_workspace_bytes_xcorr(count, n, n);
    if (ws_bytes) CHECK_HIP(hipMalloc(&ws, ws_bytes));
    CHECK_SPX(spx_gen_gaussian_pairs_f32(20261003ull, 0, count, n, 4.0f, 6.0f, 3.0f, ref, img, truth, NULL));
    CHECK_SPX(spx_xcorr_refine_f32(ref, img, count, n, n, upsample, SPX_CC, out, status, ws, ws_bytes, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    double* h_out = (double*)malloc(count * 2 * sizeof(double));
    double* h_truth = (double*)malloc(count * 2 * sizeof(double));
    int32_t* h_st = (int32_t*)malloc(count * sizeof(int32_t));
    CHECK_HIP(hipMemcpy(h_out, out, count * 2 * sizeof(double), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_truth, truth, count * 2 * sizeof(double), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_st, status, count * sizeof(int32_t), hipMemcpyDeviceToHost));
    double worst = 0.0;
    int bad = 0;
    for (int64_t i = 0; i < 2 * count; ++i) {
        const double d = fabs(h_out[i] - h_truth[i]);
        if (d > worst) worst = d;
    }
    for (int64_t i = 0; i < count; ++i) bad += h_st[i] != SPX_ST_OK;
    printf("pairs %dx%d upsample %d x %lld: max |shift - truth| %.2e px, %d non-OK\n", n, n, upsample,
           (long long)count, worst, bad);
    free(h_out); free(h_truth); free(h_st);
    (void)hipFree(ref); (void)hipFree(img); (void)hipFree(truth); (void)hipFree(out); (void)hipFree(status);
    if (ws) (void)hipFree(ws);
    return (worst < tol && bad == 0) ? 0 : 1;
}

/* reference mode over two sources of different shapes in one launch: the four "dithers" are copies
 * of the reference cutout displaced by whole pixels, so the expected displacement is exact */
static int run_var(void) {
    const int shp[4] = {40, 56, 64, 33};                    /* (ny, nx) of the two items */
    const int64_t off[2] = {0, 40 * 56};
    const size_t total = 40 * 56 + 64 * 33;
    float* h_ref = (float*)calloc(total, sizeof(float));
    float* h_im4 = (float*)calloc(4 * total, sizeof(float));
    for (int k = 0; k < 2; ++k) {
        const int ny = shp[2 * k], nx = shp[2 * k + 1];
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                const double r2 = (x - nx / 2) * (x - nx / 2) + (y - ny / 2) * (y - ny / 2);
                const double r2s = (x - nx / 2 - 2) * (x - nx / 2 - 2) + (y - ny / 2 + 1) * (y - ny / 2 + 1);
                h_ref[off[k] + (size_t)y * nx + x] = (float)exp(-r2 / 18.0);
                for (int q = 0; q < 4; ++q)                 /* all four dithers identical: shift (+2, -1) */
                    h_im4[4 * off[k] + (size_t)q * ny * nx + (size_t)y * nx + x] = (float)exp(-r2s / 18.0);
            }
    }
    float *ref, *im4, *icc;
    int64_t* d_off;
    int32_t *d_shp, *status;
    double* out;
    CHECK_HIP(hipMalloc((void**)&ref, total * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&im4, 4 * total * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&icc, 4 * total * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&d_off, sizeof(off)));
    CHECK_HIP(hipMalloc((void**)&d_shp, sizeof(shp)));
    CHECK_HIP(hipMalloc((void**)&status, 2 * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void**)&out, 4 * sizeof(double)));
    CHECK_HIP(hipMemcpy(ref, h_ref, total * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(im4, h_im4, 4 * total * sizeof(float), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_off, off, sizeof(off), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_shp, shp, sizeof(shp), hipMemcpyHostToDevice));
    CHECK_SPX(spx_find_displacement5_var_f32(ref, im4, d_off, d_shp, 2, 64, SPX_NCC, out, status, icc, NULL, 0, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    double h_out[4];
    int32_t h_st[2];
    CHECK_HIP(hipMemcpy(h_out, out, sizeof(h_out), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(h_st, status, sizeof(h_st), hipMemcpyDeviceToHost));
    printf("variable shapes: (%.4f, %.4f) status %d, (%.4f, %.4f) status %d\n", h_out[0], h_out[1], h_st[0],
           h_out[2], h_out[3], h_st[1]);
    int ok = 1;
    /* identical dithers make the interlaced image blocky: the fitted peak sits a quarter pixel below the
     * integer shift in both axes, exactly as the reference's own arithmetic gives (0.5 xm - xc) */
    for (int k = 0; k < 2; ++k)
        ok &= fabs(h_out[2 * k] - 2.0) < 0.3 && fabs(h_out[2 * k + 1] + 1.0) < 0.3 && h_st[k] == SPX_ST_OK;
    free(h_ref); free(h_im4);
    (void)hipFree(ref); (void)hipFree(im4); (void)hipFree(icc); (void)hipFree(d_off); (void)hipFree(d_shp);
    (void)hipFree(status); (void)hipFree(out);
    return ok ? 0 : 1;
}

int main(void) {
    if (spx_abi_version() != SPX_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 4; }
    if (spx_device_count() < 1) { fprintf(stderr, "no device\n"); return 5; }
    CHECK_SPX(spx_init(0));
    CHECK_SPX(spx_prepare(10));
    int rc = 0;
    rc |= run_pairs(64, 10, 20000, 1e-3);        /* 64 tile */
    rc |= run_pairs(80, 10, 2000, 1e-3);         /* fold path */
    rc |= run_pairs(128, 20, 2000, 1e-3);        /* period 192 (workspace) */
    rc |= run_var();
    /* argument errors are reported, not crashed on */
    if (spx_xcorr_refine_f32(NULL, NULL, 1, 64, 64, 10, SPX_CC, NULL, NULL, NULL, 0, NULL) != SPX_E_ARG) rc |= 1;
    if (strlen(spx_last_error()) == 0) rc |= 1;
    CHECK_SPX(spx_shutdown());
    printf(rc ? "FAILED\n" : "C ABI smoke OK\n");
    return rc;
}
