// spx_aux_kernels.h -- kernels either side of the cross-correlation hot path:
//   find_peak_kernel      centroid.find_peak in full generality (centroid.py:18-236)
//   gather_cutouts_kernel frame -> fixed tiles (cutout.py:737-755, align.py:661)
//   gen_pairs_kernel      synthetic Gaussian-spot pairs for bench / tests
//   blot_affine4_kernel   the four half-pixel dithered blots of align.py:664-676 (poly5), affine map
//   blot_poly4_kernel     the same for a polynomial (distorted) coordinate map, degree <= 5
// Needs spx_rt_hip.h (or the CPU harness) and spx_kernels.h first.
#pragma once

namespace spx {

constexpr int kPeakMaxFitPoints = 1024;

// ---------------------------------------------------------------------------
// counter-based generator: splitmix64 stream keyed by (seed, pair index).
// subpixal_amd/synth.py implements the same arithmetic with numpy uint64.
// ---------------------------------------------------------------------------
SPX_DEVICE uint64_t splitmix_next(uint64_t& z) {
    z += 0x9E3779B97F4A7C15ull;
    uint64_t r = z;
    r = (r ^ (r >> 30)) * 0xBF58476D1CE4E5B9ull;
    r = (r ^ (r >> 27)) * 0x94D049BB133111EBull;
    return r ^ (r >> 31);
}
SPX_DEVICE double splitmix_uniform(uint64_t& z) {
    return (double)(splitmix_next(z) >> 11) * (1.0 / 9007199254740992.0);
}

SPX_TKERNEL(256)
void gen_pairs_kernel(uint64_t seed, int64_t first_index, int64_t nbatch, int n, float sigma_lo,
                      float sigma_hi, float max_shift, float* __restrict__ ref,
                      float* __restrict__ img, double* __restrict__ truth) {
    const int tid = rt::thread_id();
    const int64_t npx = (int64_t)n * n;
    for (int64_t p = rt::block_id(); p < nbatch; p += rt::grid_size()) {
        uint64_t z = seed + (uint64_t)(first_index + p + 1) * 0xD1342543DE82EF95ull;
        const double tx = (2.0 * splitmix_uniform(z) - 1.0) * (double)max_shift;
        const double ty = (2.0 * splitmix_uniform(z) - 1.0) * (double)max_shift;
        const double sg = (double)sigma_lo + splitmix_uniform(z) * (double)(sigma_hi - sigma_lo);
        const double am = 0.5 + 1.5 * splitmix_uniform(z);
        const float c = 0.5f * (float)(n - 1);
        const float cx = c + (float)tx, cy = c + (float)ty;
        const float inv = -0.5f / ((float)sg * (float)sg);
        const float amp = (float)am;
        for (int64_t i = tid; i < npx; i += 256) {
            const float y = (float)(i / n), x = (float)(i % n);
            const float r2 = (x - c) * (x - c) + (y - c) * (y - c);
            const float i2 = (x - cx) * (x - cx) + (y - cy) * (y - cy);
            ref[p * npx + i] = amp * __builtin_expf(r2 * inv);
            img[p * npx + i] = amp * __builtin_expf(i2 * inv);
        }
        if (truth && tid == 0) {
            truth[2 * p] = tx;
            truth[2 * p + 1] = ty;
        }
    }
}

// ---------------------------------------------------------------------------
// cutout packing.  boxes[b] = (x0, y0, w, h) in frame pixels.  Inside the box:
// the frame pixel, or `fill` where the box overhangs the frame, the pixel is
// flagged in fmask, or it is not finite (cutout.py:737-755 + align.py:661).
// Outside the box (tile padding): 0, which leaves the linear cross-correlation
// unchanged.
// ---------------------------------------------------------------------------
// seg/ids (optional): the segmentation image and the label of each box's source; pixels of
// other labels are written as `fill` too (cutout.py:190: mask |= ~(seg == sid), then
// align.py:661 zeroes the masked pixels).
SPX_TKERNEL(256)
void gather_cutouts_kernel(const float* __restrict__ frame, const uint8_t* __restrict__ fmask,
                           int fny, int fnx, const int32_t* __restrict__ boxes, int64_t nbatch,
                           int tny, int tnx, float fill, float* __restrict__ tiles,
                           const int32_t* __restrict__ seg, const int32_t* __restrict__ ids) {
    const int64_t total = nbatch * tny * tnx;
    const int64_t step = rt::grid_size() * 256;
    for (int64_t i = rt::block_id() * 256 + rt::thread_id(); i < total; i += step) {
        const int tx = (int)(i % tnx);
        const int ty = (int)((i / tnx) % tny);
        const int64_t b = i / ((int64_t)tnx * tny);
        const int x0 = boxes[4 * b], y0 = boxes[4 * b + 1];
        const int w = boxes[4 * b + 2], h = boxes[4 * b + 3];
        float v = 0.0f;
        if (tx < w && ty < h) {
            const int fx = x0 + tx, fy = y0 + ty;
            v = fill;
            if (fx >= 0 && fx < fnx && fy >= 0 && fy < fny) {
                const float f = frame[(int64_t)fy * fnx + fx];
                const bool bad = (fmask && fmask[(int64_t)fy * fnx + fx]) || !(f - f == 0.0f) ||
                                 (seg && seg[(int64_t)fy * fnx + fx] != ids[b]);
                if (!bad) v = f;
            }
        }
        tiles[i] = v;
    }
}

// ---------------------------------------------------------------------------
// Half-pixel dithered blots: the four `blot_cutout(dzct, imct)` calls of
// align.py:664-676 for a batch of sources in one launch, for coordinate maps that are
// affine over a cutout.  The reference resamples through drizzlepac's C `tblot` with
// interp='poly5' (blot.py:79, 140-146); drizzlepac is not in the reference tree, so this
// restates the published algorithm: the quintic of IRAF's bipoly5 interpolant in Everett's
// central-difference form,
//   f(x0 + s) = t [f0 + (t^2-1)/6 d2f0 + (t^2-1)(t^2-4)/120 d4f0]
//             + s [f1 + (s^2-1)/6 d2f1 + (s^2-1)(s^2-4)/120 d4f1],   t = 1 - s,
// applied along x to six rows and then along y; samples outside the source are continued
// by v(-k) = 2 v(0) - v(k) (and likewise at the far edge); points that map outside the
// source give 0 (tblot's misval, blot.py:113).  Parity with drizzlepac is UNPINNED.
//   dither q = 00, 10, 01, 11 <-> (ox, oy) in {0, 1/2}^2: imct.dx -= 0.5 puts cutout pixel x at
//   image position x + blc - dx0 + 1/2 (cutout.py:1138), so target pixel (x, y) samples the
//   source at (xs, ys) = A (x + ox, y + oy) + b,  affine = (a0..a5): xs = a0 x' + a1 y' + a2.
// ---------------------------------------------------------------------------
SPX_DEVICE float everett5(const float (&c)[6], float s) {
    const float t = 1.0f - s, s2 = s * s, t2 = t * t;
    const float cd20 = (1.0f / 6.0f) * (c[3] - 2.0f * c[2] + c[1]);
    const float cd21 = (1.0f / 6.0f) * (c[4] - 2.0f * c[3] + c[2]);
    const float cd40 = (1.0f / 120.0f) * (c[0] - 4.0f * c[1] + 6.0f * c[2] - 4.0f * c[3] + c[4]);
    const float cd41 = (1.0f / 120.0f) * (c[1] - 4.0f * c[2] + 6.0f * c[3] - 4.0f * c[4] + c[5]);
    return s * (c[3] + (s2 - 1.0f) * (cd21 + (s2 - 4.0f) * cd41)) +
           t * (c[2] + (t2 - 1.0f) * (cd20 + (t2 - 4.0f) * cd40));
}
// sample (j, i) of an [ny][nx] tile with the edge continuation described above
SPX_DEVICE float blot_sample(const float* __restrict__ src, int ny, int nx, int j, int i) {
    // continuation index and weights: v(idx) = a v(edge) + b v(mirror)
    int ie = i, im = i, je = j, jm = j;
    float ai = 0.0f, bi = 1.0f, aj = 0.0f, bj = 1.0f;
    if (i < 0) { ie = 0; im = -i; ai = 2.0f; bi = -1.0f; }
    else if (i > nx - 1) { ie = nx - 1; im = 2 * (nx - 1) - i; ai = 2.0f; bi = -1.0f; }
    if (j < 0) { je = 0; jm = -j; aj = 2.0f; bj = -1.0f; }
    else if (j > ny - 1) { je = ny - 1; jm = 2 * (ny - 1) - j; aj = 2.0f; bj = -1.0f; }
    const float vee = src[(int64_t)je * nx + ie], vem = src[(int64_t)je * nx + im];
    const float vme = src[(int64_t)jm * nx + ie], vmm = src[(int64_t)jm * nx + im];
    const float rowe = ai * vee + bi * vem;     // row je, continued along x
    const float rowm = ai * vme + bi * vmm;     // row jm, continued along x
    return aj * rowe + bj * rowm;
}

// value of the tile at source position (xs, ys): separable quintic through the six nearest
// samples per axis; 0 outside the tile (tblot's `misval`)
SPX_DEVICE float blot_resample(const float* __restrict__ tile, int sny, int snx, double xs, double ys) {
    if (!(xs >= 0.0 && xs <= (double)(snx - 1) && ys >= 0.0 && ys <= (double)(sny - 1))) return 0.0f;
    const int ix = (int)xs, iy = (int)ys;    // floor: both are >= 0
    const float sx = (float)(xs - (double)ix), sy = (float)(ys - (double)iy);
    float col[6];
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) {
        float c[6];
        const int j = iy - 2 + jj;
        const bool inner = j >= 0 && j < sny && ix >= 2 && ix + 3 < snx;
#pragma unroll
        for (int ii = 0; ii < 6; ++ii)
            c[ii] = inner ? tile[(int64_t)j * snx + ix - 2 + ii]
                          : blot_sample(tile, sny, snx, j, ix - 2 + ii);
        col[jj] = everett5(c, sx);
    }
    return everett5(col, sy);
}

SPX_TKERNEL(256)
void blot_affine4_kernel(const float* __restrict__ src, int64_t nbatch, int sny, int snx,
                         const double* __restrict__ affine, const float* __restrict__ gain,
                         int ny, int nx, float* __restrict__ im4) {
    const int64_t per = (int64_t)4 * ny * nx;
    const int64_t total = nbatch * per;
    const int64_t step = rt::grid_size() * 256;
    for (int64_t g = rt::block_id() * 256 + rt::thread_id(); g < total; g += step) {
        const int64_t b = g / per;
        const int r = (int)(g - b * per);
        const int q = r / (ny * nx);                 // 0: 00, 1: 10, 2: 01, 3: 11
        const int y = (r - q * ny * nx) / nx, x = r % nx;
        const double ox = (q & 1) ? 0.5 : 0.0, oy = (q & 2) ? 0.5 : 0.0;
        const double* a = affine + 6 * b;
        const double xt = (double)x + ox, yt = (double)y + oy;
        const double xs = a[0] * xt + a[1] * yt + a[2];
        const double ys = a[3] * xt + a[4] * yt + a[5];
        float v = blot_resample(src + b * (int64_t)sny * snx, sny, snx, xs, ys);
        if (gain) v *= gain[b];
        im4[g] = v;
    }
}

// ---------------------------------------------------------------------------
// The same four blots for a coordinate map that is NOT affine over the cutout (instrument
// distortion: what BlotWCSMap evaluates per pixel at blot.py:71-76): per source a bivariate
// polynomial of total degree `degree` <= 5 in the target position relative to the cutout centre,
//   u = x + ox - (nx-1)/2,  v = y + oy - (ny-1)/2,
//   xs = sum_{d=0..degree} sum_{i=d..0} cx[k] u^i v^(d-i),   k = d (d+1)/2 + (d - i)
// (coef[b][0][k] for xs, coef[b][1][k] for ys; 21 slots per axis whatever the degree), evaluated
// in float64 by Horner's rule in u per power of v.
// ---------------------------------------------------------------------------
constexpr int kBlotPolyTerms = 21;
SPX_DEVICE double blot_poly_eval(const double* __restrict__ c, int degree, double u, double v) {
    // sum over j (power of v) of v^j * (sum over i of c[i, j] u^i), i + j <= degree
    double acc = 0.0, vj = 1.0;
    for (int j = 0; j <= degree; ++j) {
        double inner = 0.0;
        for (int i = degree - j; i >= 0; --i) {
            const int d = i + j;
            inner = inner * u + c[d * (d + 1) / 2 + j];
        }
        acc += vj * inner;
        vj *= v;
    }
    return acc;
}

SPX_TKERNEL(256)
void blot_poly4_kernel(const float* __restrict__ src, int64_t nbatch, int sny, int snx,
                       const double* __restrict__ coef, int degree, const float* __restrict__ gain,
                       int ny, int nx, float* __restrict__ im4) {
    const int64_t per = (int64_t)4 * ny * nx;
    const int64_t total = nbatch * per;
    const int64_t step = rt::grid_size() * 256;
    const double xc = 0.5 * (double)(nx - 1), yc = 0.5 * (double)(ny - 1);
    for (int64_t g = rt::block_id() * 256 + rt::thread_id(); g < total; g += step) {
        const int64_t b = g / per;
        const int r = (int)(g - b * per);
        const int q = r / (ny * nx);                 // 0: 00, 1: 10, 2: 01, 3: 11
        const int y = (r - q * ny * nx) / nx, x = r % nx;
        const double u = (double)x + ((q & 1) ? 0.5 : 0.0) - xc;
        const double v = (double)y + ((q & 2) ? 0.5 : 0.0) - yc;
        const double* c = coef + (int64_t)b * 2 * kBlotPolyTerms;
        const double xs = blot_poly_eval(c, degree, u, v);
        const double ys = blot_poly_eval(c + kBlotPolyTerms, degree, u, v);
        float val = blot_resample(src + b * (int64_t)sny * snx, sny, snx, xs, ys);
        if (gain) val *= gain[b];
        im4[g] = val;
    }
}

// ---------------------------------------------------------------------------
// Catalog-scale packing (round 3): the same two operations for cutouts of DIFFERENT shapes -- the reference's
// cutouts are bounding boxes + padding, one shape per source (cutout.py:159-175) -- writing the packed
// layouts the variable-shape reference-mode kernels read (spx_kernels.h ItemTable): item p's pixels at
// element off[p] of the output, row-major (h x w), its four blots at 4 off[p].  One workgroup per item
// (grid-stride), so a 5000-source catalog is one launch and no host loop touches a pixel.
// ---------------------------------------------------------------------------
SPX_TKERNEL(256)
void gather_cutouts_var_kernel(const float* __restrict__ frame, const uint8_t* __restrict__ fmask,
                               int fny, int fnx, const int32_t* __restrict__ boxes, int64_t nbatch,
                               const int64_t* __restrict__ off, float fill, float* __restrict__ out,
                               const int32_t* __restrict__ seg, const int32_t* __restrict__ ids) {
    for (int64_t b = rt::block_id(); b < nbatch; b += rt::grid_size()) {
        const int x0 = boxes[4 * b], y0 = boxes[4 * b + 1];
        const int w = boxes[4 * b + 2], h = boxes[4 * b + 3];
        float* dst = out + off[b];
        const int npx = w * h;
        for (int i = rt::thread_id(); i < npx; i += 256) {
            const int ty = i / w, tx = i - ty * w;
            const int fx = x0 + tx, fy = y0 + ty;
            float v = fill;
            if (fx >= 0 && fx < fnx && fy >= 0 && fy < fny) {
                const float f = frame[(int64_t)fy * fnx + fx];
                const bool bad = (fmask && fmask[(int64_t)fy * fnx + fx]) || !(f - f == 0.0f) ||
                                 (seg && seg[(int64_t)fy * fnx + fx] != ids[b]);
                if (!bad) v = f;
            }
            dst[i] = v;
        }
    }
}

// the four dithered blots of every source from its own (variable-shape) drizzled cutout: `map` holds 6
// doubles per source (degree 0: affine, as blot_affine4_kernel) or 2 x 21 (degree 1..5: polynomial, as
// blot_poly4_kernel); the same arithmetic per output pixel as those kernels
SPX_TKERNEL(256)
void blot4_var_kernel(const float* __restrict__ src, const int64_t* __restrict__ src_off,
                      const int32_t* __restrict__ src_shp, int64_t nbatch, const double* __restrict__ map,
                      int degree, const float* __restrict__ gain, const int64_t* __restrict__ dst_off,
                      const int32_t* __restrict__ dst_shp, float* __restrict__ im4) {
    for (int64_t b = rt::block_id(); b < nbatch; b += rt::grid_size()) {
        const int sny = src_shp[2 * b], snx = src_shp[2 * b + 1];
        const int ny = dst_shp[2 * b], nx = dst_shp[2 * b + 1];
        const float* tile = src + src_off[b];
        float* dst = im4 + 4 * dst_off[b];
        const int npx = ny * nx;
        const float g = gain ? gain[b] : 1.0f;
        if (sny < 6 || snx < 6) {            // (the fixed-shape entry refuses such sources; here: no signal)
            for (int i = rt::thread_id(); i < 4 * npx; i += 256) dst[i] = 0.0f;
            continue;
        }
        const double xc = 0.5 * (double)(nx - 1), yc = 0.5 * (double)(ny - 1);
        for (int r = rt::thread_id(); r < 4 * npx; r += 256) {
            const int q = r / npx;                       // 0: 00, 1: 10, 2: 01, 3: 11
            const int y = (r - q * npx) / nx, x = r % nx;
            const double xt = (double)x + ((q & 1) ? 0.5 : 0.0), yt = (double)y + ((q & 2) ? 0.5 : 0.0);
            double xs, ys;
            if (degree == 0) {
                const double* a = map + 6 * b;
                xs = a[0] * xt + a[1] * yt + a[2];
                ys = a[3] * xt + a[4] * yt + a[5];
            } else {
                const double* c = map + (int64_t)b * 2 * kBlotPolyTerms;
                xs = blot_poly_eval(c, degree, xt - xc, yt - yc);
                ys = blot_poly_eval(c + kBlotPolyTerms, degree, xt - xc, yt - yc);
            }
            float v = blot_resample(tile, sny, snx, xs, ys);
            if (gain) v *= g;
            dst[r] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// Bounding boxes of all segments of a label image in ONE pass (the reference scans the
// whole frame once per source: `segmentation_image == sid` + np.where, cutout.py:151-160,
// O(N_src * N_pix)).  boxes[l] = (xmin, ymin, xmax, ymax), counts[l] = pixels, for labels
// 1..max_label; empty labels keep (INT_MAX, INT_MAX, -1, -1, 0).  HBM-bound integer work:
// 4 bytes read per pixel, atomics only for labelled pixels (runs of equal labels inside a
// thread's 4 pixels are merged first).
// ---------------------------------------------------------------------------
SPX_TKERNEL(256)
void label_bbox_init_kernel(int32_t* __restrict__ boxes, int32_t* __restrict__ counts, int nlabels) {
    const int64_t step = rt::grid_size() * 256;
    for (int64_t i = rt::block_id() * 256 + rt::thread_id(); i < nlabels; i += step) {
        boxes[4 * i] = 0x7fffffff;
        boxes[4 * i + 1] = 0x7fffffff;
        boxes[4 * i + 2] = -1;
        boxes[4 * i + 3] = -1;
        counts[i] = 0;
    }
}

SPX_DEVICE void label_bbox_flush(int32_t* boxes, int32_t* counts, int label, int max_label, int y,
                                 int x1, int x2) {
    if (label <= 0 || label > max_label) return;
    rt::atomic_min_i32(boxes + 4 * (int64_t)label, x1);
    rt::atomic_min_i32(boxes + 4 * (int64_t)label + 1, y);
    rt::atomic_max_i32(boxes + 4 * (int64_t)label + 2, x2);
    rt::atomic_max_i32(boxes + 4 * (int64_t)label + 3, y);
    rt::atomic_add_i32(counts + label, x2 - x1 + 1);
}

SPX_TKERNEL(256)
void label_bbox_kernel(const int32_t* __restrict__ seg, int fny, int fnx, int max_label,
                       int32_t* __restrict__ boxes, int32_t* __restrict__ counts) {
    const int chunks = (fnx + 3) / 4;                       // 4-pixel chunks per row
    const int64_t total = (int64_t)fny * chunks;
    const int64_t step = rt::grid_size() * 256;
    for (int64_t i = rt::block_id() * 256 + rt::thread_id(); i < total; i += step) {
        const int y = (int)(i / chunks);
        const int x0 = (int)(i - (int64_t)y * chunks) * 4;
        int run_label = 0, run_x1 = 0, run_x2 = 0;
        for (int e = 0; e < 4; ++e) {
            const int x = x0 + e;
            const int l = x < fnx ? seg[(int64_t)y * fnx + x] : 0;
            if (l == run_label) { run_x2 = x; continue; }
            label_bbox_flush(boxes, counts, run_label, max_label, y, run_x1, run_x2);
            run_label = l; run_x1 = x; run_x2 = x;
        }
        label_bbox_flush(boxes, counts, run_label, max_label, y, run_x1, run_x2);
    }
}

// ---------------------------------------------------------------------------
// general find_peak
// ---------------------------------------------------------------------------
SPX_DEVICE bool better_d(double v, int i, double bv, int bi) {
    return (v > bv) || (v == bv && i < bi);
}

// workgroup arg-max of (double, int); red_d/red_i: LDS double[4]/int[4]
SPX_DEVICE void block_argmax_d(double* red_d, int* red_i, double& v, int& idx) {
    const int tid = rt::thread_id();
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double ov = rt::shfl_xor(v, m);
        const int oi = rt::shfl_xor(idx, m);
        if (better_d(ov, oi, v, idx)) { v = ov; idx = oi; }
    }
    if ((tid & 63) == 0) { red_d[tid >> 6] = v; red_i[tid >> 6] = idx; }
    rt::block_sync();
    v = red_d[0];
    idx = red_i[0];
    for (int w = 1; w < kThreads / 64; ++w)
        if (better_d(red_d[w], red_i[w], v, idx)) { v = red_d[w]; idx = red_i[w]; }
    rt::block_sync();
}

// utils.py:144-161
SPX_DEVICE double py2round(double x) { return x >= 0.0 ? floor(x + 0.5) : ceil(x - 0.5); }

SPX_DEVICE int imin(int a, int b) { return a < b ? a : b; }
SPX_DEVICE int imax2(int a, int b) { return a > b ? a : b; }

// Minimum-norm least squares  min |A c - d|  for the N x 6 design matrix held
// column-major in LDS (acol[j*N + i]), by one-sided (Hestenes) Jacobi SVD, run by
// wave 0.  Singular values <= eps*max(N,6)*s_max are dropped, as
// numpy.linalg.lstsq(rcond=None) does at centroid.py:207 (this reproduces its
// minimum-norm answers on rank-deficient boxes, e.g. a 2x3 fit box).
// vmat: LDS double[36]; coef: LDS double[6] (result).
SPX_DEVICE void jacobi_lstsq_wave0(double* acol, const double* dvec, int N, double* vmat,
                                   double* coef) {
    const int lane = rt::thread_id() & 63;
    if (lane < 6)
        for (int j = 0; j < 6; ++j) vmat[lane * 6 + j] = (lane == j) ? 1.0 : 0.0;
    rt::wave_sync();
    for (int sweep = 0; sweep < 30; ++sweep) {
        int rotations = 0;
        for (int p = 0; p < 5; ++p) {
            for (int q = p + 1; q < 6; ++q) {
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int i = lane; i < N; i += 64) {
                    const double ap = acol[p * N + i], aq = acol[q * N + i];
                    al += ap * ap;
                    be += aq * aq;
                    ga += ap * aq;
                }
                al = wave_sum(al);
                be = wave_sum(be);
                ga = wave_sum(ga);
                if (ga == 0.0 || fabs(ga) <= 1e-15 * sqrt(al * be)) continue;
                ++rotations;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta == 0.0)
                                     ? 1.0
                                     : (zeta > 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = lane; i < N; i += 64) {
                    const double ap = acol[p * N + i], aq = acol[q * N + i];
                    acol[p * N + i] = c * ap - s * aq;
                    acol[q * N + i] = s * ap + c * aq;
                }
                if (lane < 6) {
                    const double vp = vmat[lane * 6 + p], vq = vmat[lane * 6 + q];
                    vmat[lane * 6 + p] = c * vp - s * vq;
                    vmat[lane * 6 + q] = s * vp + c * vq;
                }
            }
        }
        if (rotations == 0) break;
    }
    rt::wave_sync();
    double sig2[6], proj[6];
    double smax2 = 0.0;
    for (int j = 0; j < 6; ++j) {
        double s2 = 0.0, pj = 0.0;
        for (int i = lane; i < N; i += 64) {
            const double a = acol[j * N + i];
            s2 += a * a;
            pj += a * dvec[i];
        }
        sig2[j] = wave_sum(s2);
        proj[j] = wave_sum(pj);
        if (sig2[j] > smax2) smax2 = sig2[j];
    }
    if (lane == 0) {
        const double rcond = 2.220446049250313e-16 * (double)(N > 6 ? N : 6);
        const double cut = rcond * sqrt(smax2);
        for (int r = 0; r < 6; ++r) {
            double x = 0.0;
            for (int j = 0; j < 6; ++j)
                if (sqrt(sig2[j]) > cut) x += vmat[r * 6 + j] * (proj[j] / sig2[j]);
            coef[r] = x;
        }
    }
    rt::wave_sync();
}

SPX_TKERNEL(256)
void find_peak_kernel(const double* __restrict__ image, const uint8_t* __restrict__ mask,
                      const double* __restrict__ guess, int64_t nbatch, int ny, int nx, int wx,
                      int wy, int sbx, int sby, double* __restrict__ out,
                      int* __restrict__ status) {
    SPX_STATIC_LDS(double, acol, 6 * kPeakMaxFitPoints);
    SPX_STATIC_LDS(double, dvec, kPeakMaxFitPoints);
    SPX_STATIC_LDS(double, small, 64);     // [0..3] reduction, [8..43] V, [48..53] coef
    SPX_STATIC_LDS(int, smalli, 16);       // [0..3] reduction, [8] point count
    const int tid = rt::thread_id();
    const int64_t npx = (int64_t)ny * nx;
    for (int64_t b = rt::block_id(); b < nbatch; b += rt::grid_size()) {
        const double* img = image + b * npx;
        const uint8_t* msk = mask ? mask + b * npx : nullptr;
        bool have_guess = guess != nullptr;
        double rx = 0.0, ry = 0.0;
        int st = ST_OK;
        for (int pass = 0; pass < 2; ++pass) {
            int imax, jmax;
            double cx, cy;
            bool expand = false;
            if (!have_guess) {                                   // centroid.py:111-127
                double bv = -__builtin_inf();
                int bi = 0x7fffffff;
                for (int64_t i = tid; i < npx; i += kThreads) {
                    if (msk && !msk[i]) continue;
                    const double v = img[i];
                    if (better_d(v, (int)i, bv, bi)) { bv = v; bi = (int)i; }
                }
                block_argmax_d(small, smalli, bv, bi);
                if (bi == 0x7fffffff) bi = 0;                    // nothing selectable
                jmax = bi / nx;
                imax = bi % nx;
                cx = (double)imax;
                cy = (double)jmax;
            } else {                                             // centroid.py:129-156
                cx = guess[2 * b];
                cy = guess[2 * b + 1];
                imax = (int)py2round(cx);
                jmax = (int)py2round(cy);
                if (sbx > 0) {
                    const int x1 = imax2(0, imax - sbx / 2), x2 = imin(nx, x1 + sbx);
                    const int y1 = imax2(0, jmax - sby / 2), y2 = imin(ny, y1 + sby);
                    if (x1 < x2 && y1 < y2) {
                        const int bw = x2 - x1, bh = y2 - y1;
                        double bv = -__builtin_inf();
                        int bi = 0x7fffffff;
                        for (int i = tid; i < bw * bh; i += kThreads) {
                            const double v = img[(int64_t)(y1 + i / bw) * nx + x1 + i % bw];
                            if (better_d(v, i, bv, bi)) { bv = v; bi = i; }
                        }
                        block_argmax_d(small, smalli, bv, bi);
                        if (bi == 0x7fffffff) bi = 0;
                        imax = x1 + bi % bw;
                        jmax = y1 + bi / bw;
                        cx = (double)imax;
                        cy = (double)jmax;
                    }
                    expand = (sbx != nx || sby != ny);
                }
            }
            rx = cx; ry = cy;
            if (wx * wy < 6) { st = ST_FEWPTS; break; }           // centroid.py:160-162
            int x1 = imax2(0, imax - wx / 2), x2 = imin(nx, x1 + wx);   // :165-168
            int y1 = imax2(0, jmax - wy / 2), y2 = imin(ny, y1 + wy);
            if (imax == x1 || imax == x2 || jmax == y1 || jmax == y2) {  // :171-172
                rx = (double)imax; ry = (double)jmax; st = ST_EDGE;
                break;
            }
            if (x2 - x1 < wx) {                                   // :175-179
                if (x1 == 0) x2 = imin(nx, x1 + wx);
                if (x2 == nx) x1 = imax2(0, x2 - wx);
            }
            if (y2 - y1 < wy) {                                   // :180-184
                if (y1 == 0) y2 = imin(ny, y1 + wy);
                if (y2 == ny) y1 = imax2(0, y2 - wy);
            }
            if ((x2 - x1) * (y2 - y1) < 6) { st = ST_FEWPTS; break; }   // :186-188
            // design matrix over the (masked) box, absolute coordinates: :191-204
            const int bw = x2 - x1, bh = y2 - y1;
            if (tid == 0) {
                int n = 0;
                for (int j = 0; j < bh; ++j)
                    for (int i = 0; i < bw; ++i) {
                        const int64_t g = (int64_t)(y1 + j) * nx + x1 + i;
                        if (msk && !msk[g]) continue;
                        dvec[n++] = img[g];
                    }
                smalli[8] = n;
            }
            rt::block_sync();
            const int npts = smalli[8];
            if (npts < 6) { st = ST_FEWPTS; rt::block_sync(); break; }
            if (tid == 0) {
                int n = 0;
                for (int j = 0; j < bh; ++j)
                    for (int i = 0; i < bw; ++i) {
                        const int64_t g = (int64_t)(y1 + j) * nx + x1 + i;
                        if (msk && !msk[g]) continue;
                        const double x = (double)(x1 + i), y = (double)(y1 + j);
                        acol[0 * npts + n] = 1.0;
                        acol[1 * npts + n] = x;
                        acol[2 * npts + n] = y;
                        acol[3 * npts + n] = x * y;
                        acol[4 * npts + n] = x * x;
                        acol[5 * npts + n] = y * y;
                        ++n;
                    }
            }
            rt::block_sync();
            if (tid < 64) jacobi_lstsq_wave0(acol, dvec, npts, small + 8, small + 48);
            rt::block_sync();
            const double c10 = small[49], c01 = small[50], c11 = small[51];
            const double c20 = small[52], c02 = small[53];
            rt::block_sync();
            const double det = 4.0 * c02 * c20 - c11 * c11;           // :217-225
            if (det <= 0.0 || ((c20 > 0.0 && c02 >= 0.0) || (c20 >= 0.0 && c02 > 0.0))) {
                if (expand) { have_guess = false; continue; }
                rx = (x1 + x2) / 2.0; ry = (y1 + y2) / 2.0; st = ST_NOMAX;
                break;
            }
            const double xm = (c01 * c11 - 2.0 * c02 * c10) / det;    // :227-228
            const double ym = (c10 * c11 - 2.0 * c01 * c20) / det;
            if (xm > 0.0 && xm < nx - 1.0 && ym > 0.0 && ym < ny - 1.0) {   // :230-236
                rx = xm; ry = ym; st = ST_OK;
                break;
            }
            if (expand) { have_guess = false; continue; }
            st = ST_OUTSIDE;
            break;
        }
        if (tid == 0) {
            out[2 * b] = rx;
            out[2 * b + 1] = ry;
            if (status) status[b] = st;
        }
        rt::block_sync();
    }
}

}  // namespace spx
