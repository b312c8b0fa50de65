/*
 * subpixal_hip.h -- C ABI of libsubpixal_hip.so: the MI355X (gfx950) replacement
 * for subpixal's per-cutout cross-correlation + sub-pixel peak refinement.
 *
 * The reference (spacetelescope/subpixal) is pure Python and has NO native/FFI
 * boundary; the calls below are what a ctypes binding placed at its hot-path
 * call sites would bind (INTEGRATION.md shows the stub):
 *
 *   spx_find_displacement5_f32  <-  cc.find_displacement(ref, im00, im10, im01,
 *                                   im11, cc_type, full_output)   subpixal/cc.py:21-95,
 *                                   called once per source at subpixal/align.py:682-685;
 *                                   here for a whole batch (the loop align.py:656-699
 *                                   carries no state between sources).  _f64: float64 cutouts.
 *   spx_find_displacement5_var_f32 / _f64
 *                               <-  the same loop over sources whose cutouts differ in shape
 *                                   (bounding box + padding per source, cutout.py:159-175), one
 *                                   launch per kernel family.
 *   spx_xcorr_refine_f32 / _f64 <-  the pair / upsample=U form of the same path that
 *                                   BASELINE.json measures (one fftconvolve, cc.py:114,
 *                                   + find_peak, cc.py:86, on a U-times finer grid).
 *   spx_find_peak_f64           <-  centroid.find_peak(image, xmax, ymax, peak_fit_box,
 *                                   peak_search_box, mask)        subpixal/centroid.py:18-236.
 *   spx_gather_cutouts_f32      <-  Cutout.__init__ slicing/fill  subpixal/cutout.py:737-755
 *                                   + masked-pixel zeroing        subpixal/align.py:661.
 *   spx_label_bboxes_i32        <-  per-source bounding boxes from the segmentation image
 *                                   subpixal/cutout.py:151-160 (one pass for all sources).
 *   spx_blot_affine4_f32        <-  the four blot_cutout(dzct, imct) calls per source of
 *                                   subpixal/align.py:664-676 (blot.py:79-155, drizzlepac
 *                                   tblot, interp='poly5') for coordinate maps that are
 *                                   affine over a cutout; drizzlepac is not in the reference
 *                                   tree, the quintic is restated from the published (IRAF
 *                                   bipoly5 / Everett) formula: parity UNPINNED.
 *   spx_blot_poly4_f32          <-  the same four blots through a distorted (polynomial, degree
 *                                   <= 5) coordinate map: BlotWCSMap, blot.py:21-76.
 *   spx_gen_gaussian_pairs_f32  --  synthetic workload generator (bench / tests only).
 *
 * Conventions
 *   - All array pointers are DEVICE pointers owned by the caller (e.g. torch
 *     tensors' data_ptr()), contiguous, row-major; nothing returned is owned by
 *     the library.  `stream` is a hipStream_t (NULL = default stream).  Calls
 *     enqueue work and return without synchronising.
 *   - Return value 0 = OK, negative = SPX_E_* ; spx_last_error() gives the text
 *     (thread-local).  Per-item degenerate cases (edge peak, no maximum ...) are
 *     NOT call failures: they are reported in `out_status` (SPX_ST_*), exactly
 *     the early returns of centroid.py:171-172, 218-225, 230-236.
 *   - Constant tables (twiddles, interpolation kernels) are built on first use
 *     per device and upsampling factor; call spx_prepare() up front if the launch
 *     must not allocate (stream capture).  spx_shutdown() frees them.
 *   - Host threads may share a device: enqueues on one device are serialised by the library
 *     (a per-device mutex held only while a launch is issued).
 *   - Dynamic range: the transforms are float32.  The two cutouts of a pair may differ in amplitude
 *     by any factor up to 2^+-100 (they are balanced by an exact power of two before the product), but
 *     in plain CC mode each must keep sum(|pixel|) below ~6e18 (the zero-frequency term of the packed
 *     spectrum, sum(ref) + i sum(img), is squared in float32; NCC/ZNCC normalise first); beyond that
 *     the correlation overflows and the item comes back as SPX_ST_NONFINITE rather than as a wrong
 *     shift.
 *   - Input element types: the _f32 entries take float32 pixels (the type BASELINE.json
 *     measures); the _f64 entries take float64 pixels and form the `!= 0` masks, the pooled
 *     statistics and the normalised pixels of cc.py:131-156 in float64 -- the dtype the
 *     reference computes in for float64 cutouts -- before rounding to float32 for the
 *     transforms.  All arithmetic after staging is float32 (fit: float64) in both.
 */
#ifndef SUBPIXAL_HIP_H
#define SUBPIXAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPX_ABI_VERSION 4

/* cc_type (cc.py:107-111; anything else than NCC/ZNCC means plain CC) */
#define SPX_CC 0
#define SPX_NCC 1
#define SPX_ZNCC 2

/* per-item status */
#define SPX_ST_OK 0        /* quadratic vertex accepted                                  */
#define SPX_ST_EDGE 1      /* arg-max in row/column 0 -> integer peak (centroid.py:171)  */
#define SPX_ST_NOMAX 2     /* no maximum -> centre of the fit box   (centroid.py:218)    */
#define SPX_ST_OUTSIDE 3   /* vertex outside image -> integer peak  (centroid.py:230)    */
#define SPX_ST_WINDOW 4    /* fine peak not bracketed by the refinement window           */
#define SPX_ST_FEWPTS 5    /* find_peak: fewer than 6 usable points (centroid.py:160,186,202) */
#define SPX_ST_SHAPE 7     /* variable-shape batch: item outside the launched kernel family's sizes */
#define SPX_ST_NONFINITE 6 /* a NaN/Inf pixel made the correlation non-finite.  Pair mode: every lag is NaN,
                              result = integer position (0, 0), what numpy.argmax + centroid.py:171 give.
                              Reference mode: NaN ranks as the maximum as in numpy.argmax (centroid.py:114),
                              result = integer position of the FIRST NaN/Inf of the interlaced image     */

/* error codes */
#define SPX_E_ARG (-1)        /* bad argument                              */
#define SPX_E_SHAPE (-2)      /* cutout shape / upsample not supported     */
#define SPX_E_HIP (-3)        /* HIP runtime error                         */
#define SPX_E_WORKSPACE (-4)  /* workspace missing or too small            */

/* largest cutout side and upsampling factor the kernels accept */
#define SPX_MAX_SIDE 682
#define SPX_MAX_UPSAMPLE 59
#ifndef SPX_MAX_UPSAMPLE_GENERAL      /* (overridable only to MEASURE beyond it: tools/general_precision.py) */
#define SPX_MAX_UPSAMPLE_GENERAL 39 /* pair mode on cutouts above 128 px (float32 transforms: see spx_xcorr_refine_f32) */
#endif

int spx_abi_version(void);
int spx_device_count(void);
/* select `device` for this process/thread and build its constant tables */
int spx_init(int device);
/* Build, for the current device, the interpolation tables of `upsample` for EVERY kernel family
 * (cutouts up to 32 / 85 / 128 px) and raise the dynamic-LDS limit of every kernel instance a
 * later spx_xcorr_refine_* (with this upsample) or spx_find_displacement5_* call can dispatch to,
 * whatever its cutout shape and input type: after it such calls neither allocate nor touch
 * function attributes, so they may be issued inside a stream capture. */
int spx_prepare(int upsample);
/* spx_prepare(upsample) plus the tables of the general path (cutouts above 128 px: FFT period and
 * tables depend on the cutout size) for an (ny, nx) cutout. */
int spx_prepare_shape(int ny, int nx, int upsample);
/* Free the device tables of every device this process initialised (after synchronising them);
 * the next call builds them again.  No call may be in flight on another host thread. */
int spx_shutdown(void);
const char* spx_last_error(void);

/*
 * Bytes of device scratch the batched calls need (0 when none).  Cutouts up to 85 px per side
 * are processed entirely in registers/LDS (FFT period 64 up to 32 px, 128 up to 85 px); larger
 * ones (period 192, up to 128 px) keep per-workgroup class planes and the full convolution in a
 * workspace of 435 KiB per resident workgroup (independent of nbatch beyond the grid); the
 * general path (129..682 px, period 64 C with C = 4..16 classes per axis) needs
 * (4 C^2 64^2 + P (P + 4)) floats per workgroup, two workgroups per CU (one above 341 px).
 * For the reference mode `need_icc` adds room for the interlaced images when the
 * caller does not want them back (out_icc == NULL).
 */
size_t spx_workspace_bytes_xcorr(int64_t nbatch, int ny, int nx);
size_t spx_workspace_bytes_displacement5(int64_t nbatch, int ny, int nx, int need_icc);

/*
 * Pair mode.  ref, img: float32 (or float64) [nbatch][ny][nx].  For every pair: linear
 * cross-correlation on the zero-padded grid (FFT period 64 / 128 / 192 for cutouts up to
 * 32 / 85 / 128 px per side: scipy's next_fast_len(2n-1) for n = 32 and 64, above that the
 * smallest multiple of 64 that leaves the 'same' window free of circular aliasing, i.e. the
 * same numbers at every integer lag), arg-max over the flipped
 * 'same' window, U-times trigonometric upsampling around it, 5x5 quadratic fit
 * (find_peak(., 5, 'all')), shift = peak/U - (n-1)//2.
 *   out_dxdy   : float64 [nbatch][2]  (dx, dy)
 *   out_status : int32   [nbatch]     SPX_ST_* ; may be NULL
 * 5 <= ny, nx <= SPX_MAX_SIDE (cutouts above 128 px take the slow general path: the reference's
 * cutouts, bounding box + padding of a segment, have no upper bound, cutout.py:159-175);
 * 1 <= upsample <= SPX_MAX_UPSAMPLE (cutouts above 128 px: <= SPX_MAX_UPSAMPLE_GENERAL).
 *
 * Accuracy (the transforms are float32): the shift is within 1e-3 px of the float64 evaluation of the same
 * definition (oracle/subpixal_oracle.py xcorr_refine) for sources that FIT their cutout -- Gaussian sigma up to
 * min(15 px, smaller side / 6), i.e. FWHM up to 0.4 of the side and 35 px -- on 24..200 px: every noise-free pair
 * measured, at upsample up to 27 with the default refine, and at upsample 28..59 (up to 39 above 128 px) with
 * SPX_REFINE_F64 on 33..85 px (above 85 px the refine is float64 anyway); there the default (float32) refine
 * covers sigma <= 11 px up to upsample 39 and sigma <= 8 px up to 59 (spx_xcorr_refine_ex_* below).  4.6e-4 px
 * and better for sigma 4..6 px at any upsample in either form, which was also measured with 1 % noise (profiles/r03/width_precision_256.txt,
 * width_precision_other_sizes.txt, width_precision.txt, refine_precision.txt, general_precision.txt).
 * It is NOT a bound for every content: the distance grows with the width of the correlation peak times the
 * upsample factor, and sources that fill their cutout leave 1e-3 px for some pairs -- sigma 8..11 px in 32 px:
 * 1..2 of 128 at upsample >= 39; sigma 11..15 px in 48 px: 1 of 128 at upsample 20 and at 59; sigma 15..25 px in
 * 64..128 px: up to 6 % at upsample 59 (worst measured 4.6e-3 px).  Above 128 px it happens from upsample 40 on
 * for sigma 11..25 px (200 px: 9.7e-4 px at upsample 39, 2.1e-3 at 59): hence SPX_MAX_UPSAMPLE_GENERAL.
 */
int spx_xcorr_refine_f32(const float* ref, const float* img, int64_t nbatch, int ny, int nx,
                         int upsample, int cc_type, double* out_dxdy, int32_t* out_status,
                         void* workspace, size_t workspace_bytes, void* stream);
int spx_xcorr_refine_f64(const double* ref, const double* img, int64_t nbatch, int ny, int nx,
                         int upsample, int cc_type, double* out_dxdy, int32_t* out_status,
                         void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same with the arithmetic of the refine stage (the U-times upsampling around the coarse maximum,
 * upsample > 1) chosen by the caller.  The transforms are float32 either way.
 * Cutouts of 33..85 px per side (the 64 tile and its fold path) have two forms of that stage; up to 32 px it is
 * float32, above 85 px float64, whatever `refine` says.
 *   SPX_REFINE_DEFAULT  what spx_xcorr_refine_f32 / _f64 do: float32 matrix products (= SPX_REFINE_F32).
 *   SPX_REFINE_F64      float64 accumulation: closer to the definition, a wider accuracy domain, slower.
 *   SPX_REFINE_F32      float32, said explicitly.
 * Measured on one MI355X against the float64 definition (profiles/r03/refine_precision.txt, width_precision_256.txt;
 * rates: bench_64_u*_refine_f64.json next to bench.json, bench_64_u20.json; default_rule_cost.txt):
 *   - 4..6-px-sigma spots, 64 px, upsample 10 / 20 / 40: float32 5.5e-5 / 1.3e-4 / 1.3e-4 px, float64 1.2e-5 /
 *     2.3e-5 / 4.5e-5 px; float64 costs 14 % / 21 % of the pairs per second at upsample 10 / 20, 29 % at 28..43
 *     (8.8 instead of 6.2 ms per 1e5 pairs) and 40 % from 44 on (19.7 instead of 11.9 ms at 59): 336 / 512 float64
 *     MFMAs per wave instead of 80, and 2..7 / ~143 spilled registers at three / four window blocks
 *     (f64_live3_ab.txt, f64_live4_ab.txt: the form that spills less was slower);
 *   - the distance grows with the WIDTH of the spot (a flatter correlation peak on the fine grid).  Pairs beyond
 *     1e-3 px, of 256 per (size 64 | 85 px, sigma band, upsample) cell, noise-free:
 *       sigma <= 11 px   float32: 0 up to upsample 39, 0..3 at 59                        float64: 0
 *       sigma 11..15 px  float32: 0 up to upsample 27, 5..6 at 39, 25..30 at 59          float64: 0
 *       sigma 15..20 px  float32: 2..7 up to 27, 22..35 at 39, 47..89 at 59              float64: 0..3
 *       sigma 20..25 px  (85 px) float32: 15..33 up to 27, 60 at 39, 111 at 59           float64: 0..15
 *     So float32 (the default) kept every measured pair within 1e-3 px for sigma <= 11 px up to upsample 39 and
 *     for sigma <= 15 px up to upsample 27; float64 for sigma <= 15 px at every upsample.  A caller who refines
 *     wide sources on very fine grids (sigma 11..15 px at upsample >= 28) asks for SPX_REFINE_F64 and pays the
 *     29..40 %; making that the default for everybody was tried and withdrawn once its cost was measured.
 *     For spots that fill the cutout neither form holds 1e-3 px for every pair (float64: up to 6 % of the pairs
 *     at sigma 20..25 px / upsample 59, worst 4.4e-3 px; see the accuracy note above).
 * Any other value: SPX_E_ARG.
 */
#define SPX_REFINE_DEFAULT 0
#define SPX_REFINE_F64 1
#define SPX_REFINE_F32 2
int spx_xcorr_refine_ex_f32(const float* ref, const float* img, int64_t nbatch, int ny, int nx,
                            int upsample, int cc_type, int refine, double* out_dxdy, int32_t* out_status,
                            void* workspace, size_t workspace_bytes, void* stream);
int spx_xcorr_refine_ex_f64(const double* ref, const double* img, int64_t nbatch, int ny, int nx,
                            int upsample, int cc_type, int refine, double* out_dxdy, int32_t* out_status,
                            void* workspace, size_t workspace_bytes, void* stream);

/*
 * Reference mode (cc.find_displacement).  ref: float32 [nbatch][ny][nx];
 * im4: float32 [nbatch][4][ny][nx] in the order image00, image10, image01,
 * image11.  out_icc: float32 [nbatch][2ny][2nx] interlaced cross-correlation
 * images (full_output=True), or NULL; `workspace` must hold
 * spx_workspace_bytes_displacement5(nbatch, ny, nx, out_icc == NULL) bytes (may be NULL
 * when that is 0).
 * 3 <= ny, nx <= SPX_MAX_SIDE.
 */
int spx_find_displacement5_f32(const float* ref, const float* im4, int64_t nbatch, int ny,
                               int nx, int cc_type, double* out_dxdy, int32_t* out_status,
                               float* out_icc, void* workspace, size_t workspace_bytes,
                               void* stream);
int spx_find_displacement5_f64(const double* ref, const double* im4, int64_t nbatch, int ny,
                               int nx, int cc_type, double* out_dxdy, int32_t* out_status,
                               float* out_icc, void* workspace, size_t workspace_bytes,
                               void* stream);

/*
 * Reference mode over cutouts of DIFFERENT shapes in one launch (the reference's cutouts are
 * bounding boxes + padding, one shape per source: cutout.py:159-175, 1023-1031; the loop at
 * align.py:656-699 handles them one by one).  Items are packed back to back:
 *   item_offset : int64 [nbatch]     element offset of item k's reference cutout in `ref`; its four
 *                                    dithers start at 4*item_offset[k] of `im4` (image00, 10, 01, 11,
 *                                    ny*nx apart), its interlaced image at 4*item_offset[k] of `out_icc`
 *   item_shape  : int32 [nbatch][2]  (ny, nx) of item k
 *   family_side : any side length of the kernel family every item belongs to: items up to 32 px
 *                 with items up to 32 px, up to 64 with up to 64, up to 85 with up to 85, up to 128
 *                 with up to 128 (a family also takes every smaller shape; the Python layer groups
 *                 by the smallest one).  An item outside [3, family maximum] is not computed:
 *                 (NaN, NaN), status SPX_ST_SHAPE.
 *   out_icc     : required (float32, 4 * total pixels); workspace: spx_workspace_bytes_xcorr(nbatch,
 *                 family_side, family_side) bytes.
 * Cutouts above 128 px (general path: tables depend on the size) go through the uniform entry.
 */
int spx_find_displacement5_var_f32(const float* ref, const float* im4, const int64_t* item_offset,
                                   const int32_t* item_shape, int64_t nbatch, int family_side,
                                   int cc_type, double* out_dxdy, int32_t* out_status, float* out_icc,
                                   void* workspace, size_t workspace_bytes, void* stream);
int spx_find_displacement5_var_f64(const double* ref, const double* im4, const int64_t* item_offset,
                                   const int32_t* item_shape, int64_t nbatch, int family_side,
                                   int cc_type, double* out_dxdy, int32_t* out_status, float* out_icc,
                                   void* workspace, size_t workspace_bytes, void* stream);

/*
 * General find_peak (centroid.py:18-236), one image per workgroup.
 *   image : float64 [nbatch][ny][nx]
 *   mask  : uint8   [nbatch][ny][nx] (non-zero = good pixel) or NULL
 *   guess : float64 [nbatch][2] (xmax, ymax) or NULL (search the whole image)
 *   fit box (fit_wx, fit_wy) >= 1; search box (search_wx, search_wy), both 0 =
 *   no brute-force search ('off'/None); ignored when guess is NULL.
 *   out_xy : float64 [nbatch][2]; out_status int32 [nbatch] or NULL.
 */
int spx_find_peak_f64(const double* image, const uint8_t* mask, const double* guess,
                      int64_t nbatch, int ny, int nx, int fit_wx, int fit_wy, int search_wx,
                      int search_wy, double* out_xy, int32_t* out_status, void* stream);

/*
 * Cutout packing: gathers nbatch windows out of a frame into fixed tiles.
 *   frame : float32 [fny][fnx];  fmask: uint8 [fny][fnx] non-zero = bad pixel, or NULL
 *   boxes : int32 [nbatch][4] = (x0, y0, width, height), window may overhang the frame
 *   tiles : float32 [nbatch][tny][tnx]; pixels outside the window/frame, masked
 *           or non-finite are written as `fill` (cutout.py:737,755; align.py:661 uses 0)
 *   seg   : int32 [fny][fnx] segmentation image and ids: int32 [nbatch] the label of each
 *           box's source, or both NULL; with them, pixels of other labels are `fill` as well
 *           (cutout.py:190: mask |= ~(segmentation == sid)).
 */
int spx_gather_cutouts_f32(const float* frame, const uint8_t* fmask, int fny, int fnx,
                           const int32_t* boxes, int64_t nbatch, int tny, int tnx, float fill,
                           float* tiles, const int32_t* seg, const int32_t* ids, void* stream);

/*
 * Bounding boxes of every segment of a label image in one pass (replaces the per-source
 * `segmentation_image == sid` + np.where scans of cutout.py:151-160).
 *   seg    : int32 [fny][fnx], 0 = background, labels 1..max_label (others ignored)
 *   boxes  : int32 [max_label + 1][4] = (xmin, ymin, xmax, ymax), inclusive;
 *            labels without pixels get (INT32_MAX, INT32_MAX, -1, -1)
 *   counts : int32 [max_label + 1] pixels per label
 */
int spx_label_bboxes_i32(const int32_t* seg, int fny, int fnx, int32_t max_label,
                         int32_t* boxes, int32_t* counts, void* stream);

/*
 * Half-pixel dithered blots: for every source the four images image00, image10, image01,
 * image11 that spx_find_displacement5_f32 takes (align.py:664-676), resampled from the
 * source's drizzled cutout with the separable quintic ('poly5') interpolant.
 *   src    : float32 [nbatch][sny][snx] drizzled cutouts (masked pixels already 0), sny, snx >= 6
 *   affine : float64 [nbatch][6] = (a0..a5): target pixel (x', y') (0-based, un-dithered grid)
 *            lies at source pixel (a0 x' + a1 y' + a2, a3 x' + a4 y' + a5)
 *   gain   : float32 [nbatch] factor applied to the samples (blot.py:134-150: exposure-time /
 *            pixel-area scaling), or NULL for 1
 *   im4    : float32 [nbatch][4][ny][nx]; dither (ox, oy) in {0, 1/2}^2 (imct.dx -= ox,
 *            imct.dy -= oy) samples target position (x + ox, y + oy) (cutout.py:1138: cutout
 *            pixel x lies at image position x + blc - dx); points mapping outside the source are 0.
 */
int spx_blot_affine4_f32(const float* src, int64_t nbatch, int sny, int snx, const double* affine,
                         const float* gain, int ny, int nx, float* im4, void* stream);

/*
 * The same four blots through a map that is NOT affine over the cutout (instrument distortion:
 * what the reference's BlotWCSMap evaluates per pixel, blot.py:21-76): per source a bivariate
 * polynomial of total degree `degree` (1..5) in the target position relative to the cutout centre,
 *   u = x + ox - (nx-1)/2,  v = y + oy - (ny-1)/2,
 *   xs = sum_{d=0..degree} sum_{i=d..0} coef[b][0][k] u^i v^(d-i),  k = d(d+1)/2 + (d-i),
 *   ys likewise with coef[b][1][k];
 *   coef : float64 [nbatch][2][21] (21 slots per axis whatever the degree; unused ones ignored).
 * subpixal_amd.blot.poly_from_map fits the coefficients from any callable map and reports the
 * residual; everything else as spx_blot_affine4_f32.  Parity with drizzlepac: UNPINNED.
 */
int spx_blot_poly4_f32(const float* src, int64_t nbatch, int sny, int snx, const double* coef,
                       int degree, const float* gain, int ny, int nx, float* im4, void* stream);

/*
 * Catalog-scale entries (round 3): the body of the loop align.py:656-699 for a whole catalog whose cutouts
 * all have DIFFERENT shapes (bounding box + padding per source, cutout.py:159-175), without a host loop
 * over sources or pixels.  Packed layout shared by the three calls: item p's (h x w) pixels, row-major, at
 * element item_offset[p] of a flat float32 buffer; its four dithered blots at 4 * item_offset[p]
 * (order 00, 10, 01, 11); its interlaced image (2h x 2w) at 4 * item_offset[p] of `out_icc`;
 * item_shape = int32 [nbatch][2] = (h, w).
 */

/* frame -> packed cutouts (cutout.py:689-797 slicing/fill + align.py:661 zeroing; see spx_gather_cutouts_f32
 * for `fill`, `fmask`, `seg`, `ids`): boxes int32 [nbatch][4] = (x0, y0, w, h). */
int spx_gather_cutouts_var_f32(const float* frame, const uint8_t* fmask, int fny, int fnx,
                               const int32_t* boxes, int64_t nbatch, const int64_t* item_offset,
                               float fill, float* packed, const int32_t* seg, const int32_t* ids,
                               void* stream);

/* packed drizzled cutouts -> packed blots (the four blot_cutout calls of align.py:664-676 per source).
 *   map    : degree == 0: float64 [nbatch][6] affine maps as spx_blot_affine4_f32;
 *            degree 1..5: float64 [nbatch][2][21] polynomial maps as spx_blot_poly4_f32
 *   src_*  : layout of the drizzled cutouts, dst_* : layout of the image cutouts the blots are made for
 *            (sources smaller than 6 px per side give all-zero blots) */
int spx_blot4_var_f32(const float* src, const int64_t* src_offset, const int32_t* src_shape,
                      int64_t nbatch, const double* map, int degree, const float* gain,
                      const int64_t* dst_offset, const int32_t* dst_shape, float* im4, void* stream);

/* cc.find_displacement for every item of a packed catalog: ONE call launches each kernel family named in
 * `family_mask` (bit 0: larger side 3..32 px, bit 1: 33..64, bit 2: 65..85, bit 3: 86..128) over the same
 * tables; each family takes its own items and leaves the others alone.  Items no launched family takes
 * (a side below 3 or above 128 px, or a family whose bit is not set) are NOT written: pre-fill out_dxdy /
 * out_status with your "not measured" values.  workspace: spx_workspace_bytes_xcorr(nbatch, 128, 128)
 * bytes when bit 3 is set, else none.  out_icc must be given. */
#define SPX_FAMILY_32 1
#define SPX_FAMILY_64 2
#define SPX_FAMILY_85 4
#define SPX_FAMILY_128 8
int spx_find_displacement5_catalog_f32(const float* ref, const float* im4, const int64_t* item_offset,
                                       const int32_t* item_shape, int64_t nbatch, int family_mask,
                                       int cc_type, double* out_dxdy, int32_t* out_status,
                                       float* out_icc, void* workspace, size_t workspace_bytes,
                                       void* stream);

/*
 * Synthetic Gaussian-spot pairs: pair k = first_index + i has
 * tx, ty ~ U(-max_shift, max_shift), sigma ~ U(sigma_lo, sigma_hi), amplitude ~
 * U(0.5, 2) from a counter-based generator keyed by (seed, k); ref has the spot
 * at the tile centre, img at centre + (tx, ty).  truth_dxdy float64 [nbatch][2]
 * (may be NULL).
 */
int spx_gen_gaussian_pairs_f32(uint64_t seed, int64_t first_index, int64_t nbatch, int n,
                               float sigma_lo, float sigma_hi, float max_shift, float* ref,
                               float* img, double* truth_dxdy, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SUBPIXAL_HIP_H */
