// TEST INFRASTRUCTURE ONLY: runs the kernels of subpixal_amd/csrc/spx_kernels.h on
// CPU threads (spx_rt_emu.h) so tests can check their logic without a GPU.
// The harness is compiled as four objects (EMU_PART 1..4: pair mode float32, pair mode float64,
// reference mode, auxiliary kernels) so that `make -j` builds them side by side.
#ifndef EMU_PART
#error "compile with -DEMU_PART=1..4 (see subpixal_amd/csrc/Makefile)"
#endif
#include "spx_rt_emu.h"
#include "spx_kernels.h"
#include "spx_kernels8.h"
#include "spx_kernels5.h"
#if EMU_PART == 4
#include "spx_aux_kernels.h"      // plain (non-template) kernels: one object only
#endif
#include "spx_kernels128.h"
#include "spx_kernels_big.h"
#include "spx_kernels32.h"
#include "spx_tables.h"

using namespace spx;

#if EMU_PART == 4
extern "C" int emu_lds_bytes(int U) {
    int wb = host::window_blocks(U);
    return Lds<2>::total(wb > 0 ? 16 * wb : 0);
}
#endif

#if EMU_PART == 4
int64_t g_grid = 0;            // 0: one workgroup per item; else grid-stride over the batch
extern "C" void emu_set_grid(int64_t g) { g_grid = g; }
int g_refine64 = -1;           // the 64 tile's refine: -1 the product's default (float32: spx_capi.hip
                               // refine64_is_f64), 0 float32, 1 float64
extern "C" void emu_set_refine64(int v) { g_refine64 = v; }
// the kernels' workgroup -> first item mapping (spx_kernels.h), for the bijection test
extern "C" int64_t emu_first_item(int64_t b, int64_t nwg) { return first_item(b, nwg); }
#else
extern int64_t g_grid;
extern int g_refine64;
#endif

// ---------------------------------------------------------------------------
// pair mode and reference mode: the dispatch of spx_capi.hip (32 tile; 64 tile, with its fold
// path for 65..85 px; period 192 for 86..128 px), for float32 and float64 inputs
// ---------------------------------------------------------------------------
#if EMU_PART == 1 || EMU_PART == 2
// R: the refine stage's arithmetic (spx_kernels.h RefineF32 / RefineF64), each with its own table form
template <bool FOLD, typename TIn, typename R>
static int emu_pair64_as(const TIn* ref, const TIn* img, int64_t nbatch, int ny, int nx,
                         int U, int cc_type, double* out, int* status) {
    constexpr bool kF64 = sizeof(typename R::S) == 8;
    const int wb = host::window_blocks(U);
    std::vector<float> tw = host::make_twiddles(128);
    std::vector<float> kt;
    std::vector<double> ktd;
    if (wb > 0) { if (kF64) ktd = host::make_ktab_f64(128, U, 16 * wb); else kt = host::make_ktab(128, U, 16 * wb); }
    const cf* twp = reinterpret_cast<const cf*>(tw.data());
    const float* ktp = wb <= 0 ? nullptr : kF64 ? reinterpret_cast<const float*>(ktd.data()) : kt.data();
    auto run = [&](auto fn) {
        rt::launch(g_grid > 0 && g_grid < nbatch ? g_grid : nbatch, kThreads, fn, Lds<2>::total(16 * wb));
    };
    switch (wb) {
    case 0: run([&] { pair_kernel<2, 0, 0, FOLD, TIn, R>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status); }); break;
    case 1: run([&] { pair_kernel<2, 1, 0, FOLD, TIn, R>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status); }); break;
    case 2: run([&] { pair_kernel<2, 2, 0, FOLD, TIn, R>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status); }); break;
    case 3: run([&] { pair_kernel<2, 3, 0, FOLD, TIn, R>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status); }); break;
    default: run([&] { pair_kernel<2, 4, 0, FOLD, TIn, R>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status); }); break;
    }
    return 0;
}
template <bool FOLD, typename TIn>
static int emu_pair64(const TIn* ref, const TIn* img, int64_t nbatch, int ny, int nx,
                      int U, int cc_type, double* out, int* status) {
    const bool f64 = g_refine64 > 0;
    return f64 ? emu_pair64_as<FOLD, TIn, RefineF64>(ref, img, nbatch, ny, nx, U, cc_type, out, status)
               : emu_pair64_as<FOLD, TIn, RefineF32>(ref, img, nbatch, ny, nx, U, cc_type, out, status);
}

// the 64 tile on eight waves per pair (spx_kernels8.h)
template <typename TIn>
static int emu_pair8(const TIn* ref, const TIn* img, int64_t nbatch, int ny, int nx,
                     int U, int cc_type, double* out, int* status) {
    const int wb = host::window_blocks(U);
    std::vector<float> tw = host::make_twiddles(128);
    std::vector<float> kt;
    if (wb > 0) kt = host::make_ktab(128, U, 16 * wb);
    const cf* twp = reinterpret_cast<const cf*>(tw.data());
    const float* ktp = kt.empty() ? nullptr : kt.data();
    auto run = [&](auto fn) {
        rt::launch(g_grid > 0 && g_grid < nbatch ? g_grid : nbatch, w8::kT8, fn, w8::L8::total(16 * wb));
    };
    switch (wb) {
    case 0: run([&] { w8::pair8_kernel<0, 0, TIn>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status); }); break;
    case 1: run([&] { w8::pair8_kernel<1, 0, TIn>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status); }); break;
    case 2: run([&] { w8::pair8_kernel<2, 0, TIn>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status); }); break;
    case 3: run([&] { w8::pair8_kernel<3, 0, TIn>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status); }); break;
    default: run([&] { w8::pair8_kernel<4, 0, TIn>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status); }); break;
    }
    return 0;
}

template <int C, typename TIn>
static int emu_pair_big(const TIn* ref, const TIn* img, int64_t nbatch, int ny, int nx, int U,
                        int cc_type, double* out, int* status) {
    const int wb = host::window_blocks(U);
    std::vector<float> tw = host::make_twiddles(64 * C);
    std::vector<double> kt;
    if (wb > 0) kt = host::make_ktab_big_f64(64 * C, U, 16 * wb);
    const cf* twp = reinterpret_cast<const cf*>(tw.data());
    const double* ktp = kt.empty() ? nullptr : kt.data();
    const int64_t grid = g_grid > 0 && g_grid < nbatch ? g_grid : nbatch;
    std::vector<float> ws((size_t)grid * (LdsBig<C>::kWsBytes / sizeof(float)));
    float* wsp = ws.data();
    auto run = [&](auto fn) { rt::launch(grid, kThreads, fn, LdsBig<C>::total(16 * wb)); };
    switch (wb) {
    case 0: run([&] { pair128_kernel<C, 0, 0, TIn>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status, wsp); }); break;
    case 1: run([&] { pair128_kernel<C, 1, 0, TIn>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status, wsp); }); break;
    case 2: run([&] { pair128_kernel<C, 2, 0, TIn>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status, wsp); }); break;
    case 3: run([&] { pair128_kernel<C, 3, 0, TIn>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status, wsp); }); break;
    default: run([&] { pair128_kernel<C, 4, 0, TIn>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status, wsp); }); break;
    }
    return 0;
}

template <typename TIn>
static int emu_pair32_t(const TIn* ref, const TIn* img, int64_t nbatch, int ny, int nx,
                        int U, int cc_type, double* out, int* status) {
    const int wb = host::window_blocks(U);
    std::vector<float> tw = host::make_twiddles(64);
    std::vector<float> kt;
    if (wb > 0) kt = host::make_ktab32(U, 16 * wb);
    const cf* twp = reinterpret_cast<const cf*>(tw.data());
    const float* ktp = kt.empty() ? nullptr : kt.data();
    int64_t grid = (nbatch + 3) / 4;
    if (g_grid > 0 && g_grid < grid) grid = g_grid;
    auto run = [&](auto fn) { rt::launch(grid, kThreads, fn, Lds32::total(16 * (wb > 0 ? wb : 1))); };
    switch (wb) {
    case 0: run([&] { pair32_kernel<0, TIn>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status); }); break;
    case 1: run([&] { pair32_kernel<1, TIn>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status); }); break;
    case 2: run([&] { pair32_kernel<2, TIn>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status); }); break;
    case 3: run([&] { pair32_kernel<3, TIn>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status); }); break;
    default: run([&] { pair32_kernel<4, TIn>(ref, img, nbatch, ny, nx, U, cc_type, twp, ktp, out, status); }); break;
    }
    return 0;
}

// general path (cutouts above 128 px): class count C at run time
template <typename TIn>
static int emu_pair_general(const TIn* ref, const TIn* img, int64_t nbatch, int ny, int nx, int U,
                            int cc_type, double* out, int* status) {
    const int wb = host::window_blocks(U);
    const int C = big_class_count(ny, nx);
    if (C > kBigMaxC) return -1;
    std::vector<float> tw = host::make_twiddles(64 * C);
    std::vector<double> kt;
    if (wb > 0) kt = host::make_ktab_big_f64(64 * C, U, 16 * wb);
    const cf* twp = reinterpret_cast<const cf*>(tw.data());
    const double* ktp = kt.empty() ? nullptr : kt.data();
    const int64_t grid = g_grid > 0 && g_grid < nbatch ? g_grid : nbatch;
    std::vector<float> ws((size_t)grid * big_ws_floats(C));
    float* wsp = ws.data();
    auto run = [&](auto fn) { rt::launch(grid, kThreads, fn, LdsGen::total(16 * wb)); };
    switch (wb) {
    case 0: run([&] { pair_big_kernel<0, TIn>(ref, img, nbatch, ny, nx, U, cc_type, C, twp, ktp, out, status, wsp); }); break;
    case 1: run([&] { pair_big_kernel<1, TIn>(ref, img, nbatch, ny, nx, U, cc_type, C, twp, ktp, out, status, wsp); }); break;
    case 2: run([&] { pair_big_kernel<2, TIn>(ref, img, nbatch, ny, nx, U, cc_type, C, twp, ktp, out, status, wsp); }); break;
    case 3: run([&] { pair_big_kernel<3, TIn>(ref, img, nbatch, ny, nx, U, cc_type, C, twp, ktp, out, status, wsp); }); break;
    default: run([&] { pair_big_kernel<4, TIn>(ref, img, nbatch, ny, nx, U, cc_type, C, twp, ktp, out, status, wsp); }); break;
    }
    return 0;
}

// tile: 0 = the product's choice, else force 32 / 64 / 192 / 256 (period of the big path);
// 648 = the 64 tile's eight-wave kernel (spx_kernels8.h), 64 = its four-wave kernel
template <typename TIn>
static int emu_pair_t(const TIn* ref, const TIn* img, int64_t nbatch, int ny, int nx, int U,
                      int cc_type, double* out, int* status, int tile) {
    if (ny < 5 || nx < 5) return -1;
    if (host::window_blocks(U) < 0) return -2;
    const int n = ny > nx ? ny : nx;
    if (n > 128) return emu_pair_general<TIn>(ref, img, nbatch, ny, nx, U, cc_type, out, status);
    if (tile == 0) tile = n <= 32 ? 32 : (n <= 85 ? 64 : 192);
    if (tile == 32 && n <= 32) return emu_pair32_t<TIn>(ref, img, nbatch, ny, nx, U, cc_type, out, status);
    if constexpr (sizeof(TIn) == 4)       // (as the product library: float32 cutouts only)
        if (tile == 648 && n <= 64) return emu_pair8<TIn>(ref, img, nbatch, ny, nx, U, cc_type, out, status);
    if (tile == 64 && n <= 64) return emu_pair64<false, TIn>(ref, img, nbatch, ny, nx, U, cc_type, out, status);
    if (tile == 64 && n <= 85) return emu_pair64<true, TIn>(ref, img, nbatch, ny, nx, U, cc_type, out, status);
    if (tile == 192) return emu_pair_big<3, TIn>(ref, img, nbatch, ny, nx, U, cc_type, out, status);
    if (tile == 256) return emu_pair_big<4, TIn>(ref, img, nbatch, ny, nx, U, cc_type, out, status);
    return -3;
}
#if EMU_PART == 1
extern "C" int emu_pair_f32(const float* ref, const float* img, int64_t nbatch, int ny, int nx, int U,
                            int cc_type, double* out, int* status, int tile) {
    return emu_pair_t<float>(ref, img, nbatch, ny, nx, U, cc_type, out, status, tile);
}
#else
extern "C" int emu_pair_f64(const double* ref, const double* img, int64_t nbatch, int ny, int nx, int U,
                            int cc_type, double* out, int* status, int tile) {
    return emu_pair_t<double>(ref, img, nbatch, ny, nx, U, cc_type, out, status, tile);
}
#endif
#endif   // EMU_PART 1, 2

#if EMU_PART == 3
int g_disp5_packed = 1;          // 64-tile reference-mode kernel: 0 round 2's, 1 the product's rule (packed up to
                                 // 64 px: spx_capi.hip run_disp5), 2 always the packed one
extern "C" void emu_set_disp5_packed(int v) { g_disp5_packed = v; }
// off/shp: per-item offsets and shapes (variable-shape batch of the family of (ny, nx)), or nulls
template <typename TIn>
static int emu_disp5_t(const TIn* ref, const TIn* im4, int64_t nbatch, int ny, int nx,
                       int cc_type, float* icc, double* out, int* status,
                       const int64_t* off = nullptr, const int* shp = nullptr) {
    if (ny < 3 || nx < 3) return -1;
    const int n = ny > nx ? ny : nx;
    if (off && n > 128) return -1;
    const ItemTable items = {off, shp, n <= 32 ? 32 : (n <= 64 ? 64 : (n <= 85 ? 85 : 128)), 0};
    if (n > 128) {
        const int C = big_class_count(ny, nx);
        if (C > kBigMaxC) return -1;
        std::vector<float> tw = host::make_twiddles(64 * C);
        const cf* twp = reinterpret_cast<const cf*>(tw.data());
        std::vector<float> ws((size_t)nbatch * big_ws_floats(C));
        float* wsp = ws.data();
        rt::launch(nbatch, kThreads, [&] {
            disp5_big_kernel<TIn>(ref, im4, nbatch, ny, nx, cc_type, C, twp, icc, out, status, wsp);
        }, LdsGen::total(0));
        return 0;
    }
    if (n <= 32) {
        std::vector<float> tw = host::make_twiddles(64);
        const cf* twp = reinterpret_cast<const cf*>(tw.data());
        rt::launch((nbatch + 3) / 4, kThreads, [&] {
            disp5_32_kernel<TIn>(ref, im4, nbatch, ny, nx, cc_type, twp, icc, out, status, items);
        }, Lds32::total(16));
        return 0;
    }
    if (n <= 85) {
        std::vector<float> tw = host::make_twiddles(128);
        const cf* twp = reinterpret_cast<const cf*>(tw.data());
        if (g_disp5_packed == 2 || (g_disp5_packed == 1 && n <= 64)) {        // spx_kernels5.h
            if (n > 64)
                rt::launch(nbatch, kThreads,
                           [&] { p5::disp5p_kernel<true, TIn>(ref, im4, nbatch, ny, nx, cc_type, twp, icc, out, status, items); },
                           p5::L5<true>::TOTAL);
            else
                rt::launch(nbatch, kThreads,
                           [&] { p5::disp5p_kernel<false, TIn>(ref, im4, nbatch, ny, nx, cc_type, twp, icc, out, status, items); },
                           p5::L5<false>::TOTAL);
            return 0;
        }
        if (n > 64)
            rt::launch(nbatch, kThreads,
                       [&] { disp5_kernel<2, true, TIn>(ref, im4, nbatch, ny, nx, cc_type, twp, icc, out, status, items); },
                       Lds<2>::total(0));
        else
            rt::launch(nbatch, kThreads,
                       [&] { disp5_kernel<2, false, TIn>(ref, im4, nbatch, ny, nx, cc_type, twp, icc, out, status, items); },
                       Lds<2>::total(0));
        return 0;
    }
    std::vector<float> tw = host::make_twiddles(192);
    const cf* twp = reinterpret_cast<const cf*>(tw.data());
    std::vector<float> ws((size_t)nbatch * (kWs96Bytes / sizeof(float)));
    float* wsp = ws.data();
    rt::launch(nbatch, kThreads, [&] {
        disp5_128_kernel<3, TIn>(ref, im4, nbatch, ny, nx, cc_type, twp, icc, out, status, wsp, items);
    }, LdsBig<3>::total(0));
    return 0;
}
extern "C" int emu_disp5_f32(const float* ref, const float* im4, int64_t nbatch, int ny, int nx,
                             int cc_type, float* icc, double* out, int* status) {
    return emu_disp5_t<float>(ref, im4, nbatch, ny, nx, cc_type, icc, out, status);
}
extern "C" int emu_disp5_f64(const double* ref, const double* im4, int64_t nbatch, int ny, int nx,
                             int cc_type, float* icc, double* out, int* status) {
    return emu_disp5_t<double>(ref, im4, nbatch, ny, nx, cc_type, icc, out, status);
}
extern "C" int emu_disp5_var_f32(const float* ref, const float* im4, const int64_t* off, const int* shp,
                                 int64_t nbatch, int family_side, int cc_type, float* icc, double* out,
                                 int* status) {
    return emu_disp5_t<float>(ref, im4, nbatch, family_side, family_side, cc_type, icc, out, status, off, shp);
}

#endif   // EMU_PART 3

#if EMU_PART == 4
extern "C" int emu_find_peak(const double* image, const uint8_t* mask, const double* guess,
                             int64_t nbatch, int ny, int nx, int wx, int wy, int sbx, int sby,
                             double* out, int* status) {
    if ((int64_t)wx * wy > kPeakMaxFitPoints) return -2;
    rt::launch(nbatch, kThreads, [&] {
        find_peak_kernel(image, mask, guess, nbatch, ny, nx, wx, wy, sbx, sby, out, status);
    }, 0);
    return 0;
}

extern "C" int emu_gather(const float* frame, const uint8_t* fmask, int fny, int fnx,
                          const int32_t* boxes, int64_t nbatch, int tny, int tnx, float fill,
                          float* tiles, const int32_t* seg, const int32_t* ids) {
    const int64_t blocks = (nbatch * tny * tnx + 255) / 256;
    rt::launch(blocks < 8 ? blocks : 8, 256, [&] {
        gather_cutouts_kernel(frame, fmask, fny, fnx, boxes, nbatch, tny, tnx, fill, tiles, seg, ids);
    });
    return 0;
}

extern "C" int emu_gather_var(const float* frame, const uint8_t* fmask, int fny, int fnx, const int32_t* boxes,
                              int64_t nbatch, const int64_t* off, float fill, float* out, const int32_t* seg,
                              const int32_t* ids) {
    rt::launch(nbatch < 3 ? nbatch : 3, 256, [&] {
        gather_cutouts_var_kernel(frame, fmask, fny, fnx, boxes, nbatch, off, fill, out, seg, ids);
    }, 0);
    return 0;
}

extern "C" int emu_blot4_var(const float* src, const int64_t* src_off, const int32_t* src_shp, int64_t nbatch,
                             const double* map, int degree, const float* gain, const int64_t* dst_off,
                             const int32_t* dst_shp, float* im4) {
    rt::launch(nbatch < 3 ? nbatch : 3, 256, [&] {
        blot4_var_kernel(src, src_off, src_shp, nbatch, map, degree, gain, dst_off, dst_shp, im4);
    }, 0);
    return 0;
}

extern "C" int emu_gen_pairs(uint64_t seed, int64_t first, int64_t nbatch, int n, float slo,
                             float shi, float maxshift, float* ref, float* img, double* truth) {
    rt::launch(nbatch, 256, [&] {
        gen_pairs_kernel(seed, first, nbatch, n, slo, shi, maxshift, ref, img, truth);
    });
    return 0;
}

extern "C" int emu_label_bboxes(const int32_t* seg, int fny, int fnx, int max_label, int32_t* boxes,
                                int32_t* counts) {
    rt::launch(1, 256, [&] { label_bbox_init_kernel(boxes, counts, max_label + 1); }, 0);
    rt::launch(3, 256, [&] { label_bbox_kernel(seg, fny, fnx, max_label, boxes, counts); }, 0);
    return 0;
}

extern "C" int emu_blot_affine4(const float* src, int64_t nbatch, int sny, int snx,
                               const double* affine, const float* gain, int ny, int nx, float* im4) {
    if (sny < 6 || snx < 6) return -2;
    rt::launch(3, 256, [&] { blot_affine4_kernel(src, nbatch, sny, snx, affine, gain, ny, nx, im4); }, 0);
    return 0;
}

extern "C" int emu_blot_poly4(const float* src, int64_t nbatch, int sny, int snx, const double* coef,
                              int degree, const float* gain, int ny, int nx, float* im4) {
    if (sny < 6 || snx < 6 || degree < 1 || degree > 5) return -2;
    rt::launch(3, 256, [&] { blot_poly4_kernel(src, nbatch, sny, snx, coef, degree, gain, ny, nx, im4); }, 0);
    return 0;
}
#endif   // EMU_PART 4
