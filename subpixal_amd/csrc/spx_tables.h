// Host-side constant tables for the kernels in spx_kernels.h (plain C++).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace spx {
namespace host {

constexpr double kPi = 3.14159265358979323846264338327950288;

// w_P^j = exp(-2 pi i j / P), j in [0, P), as interleaved (re, im) floats
inline std::vector<float> make_twiddles(int P) {
    std::vector<float> tw(2 * (size_t)P);
    for (int j = 0; j < P; ++j) {
        const double a = -2.0 * kPi * (double)j / (double)P;
        tw[2 * j] = (float)std::cos(a);
        tw[2 * j + 1] = (float)std::sin(a);
    }
    return tw;
}

// number of 16-wide blocks of the fine window for upsampling factor U:
// W = 16*blocks >= U + 5 (the +-1/2 px uncertainty of the coarse arg-max plus the
// 5x5 fit box); 0 for U == 1; -1 when unsupported.
inline int window_blocks(int U) {
    if (U < 1) return -1;
    if (U == 1) return 0;
    const int blocks = (U + 5 + 15) / 16;
    return blocks <= 4 ? blocks : -1;
}

// Interpolation kernels of the two parity classes of the period-P (=128)
// trigonometric interpolant, sampled on the 64-lag grid of one class:
//   K_0(t) = 1/64 [1 + 2 sum_{j=1..31} cos(2 pi 2j t / P) + cos(2 pi (P/2) t / P)]
//   K_1(t) = 2/64 sum_{j odd in 1..P/2-1} cos(2 pi j t / P)
// (the Nyquist bin is split between +-P/2, as oracle.upsampled_cc does).
inline double class_kernel(int c, int P, double t) {
    double s = 0.0;
    if (c == 0) {
        s = 1.0 + std::cos(kPi * t);
        for (int k = 2; k < P / 2; k += 2) s += 2.0 * std::cos(2.0 * kPi * k * t / P);
    } else {
        for (int k = 1; k < P / 2; k += 2) s += 2.0 * std::cos(2.0 * kPi * k * t / P);
    }
    return s / (double)(P / 2);
}

// Lane-major operand tables of the MFMA fine-window stage (spx_kernels.h
// fine_window), W = 16*blocks, lane = 16 lk + lj:
//   [0][c][blk][lane][s]      = K_c( -(16 blk + lj - W/2)/U - (4 s + lk - 32) )
//   [1][c][blk][lane][4 t + r] = K_c( -(16 blk + lj - W/2)/U - (16 t + 4 lk + r - 32) )
inline std::vector<float> make_ktab(int P, int U, int W) {
    const int blocks = W / 16;
    std::vector<float> k((size_t)2 * 2 * blocks * 64 * 16);
    for (int which = 0; which < 2; ++which)
        for (int c = 0; c < 2; ++c)
            for (int blk = 0; blk < blocks; ++blk)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 16; ++i) {
                        const int lk = lane >> 4, lj = lane & 15;
                        const int m = which == 0 ? (4 * i + lk - 32)
                                                 : (16 * (i >> 2) + 4 * lk + (i & 3) - 32);
                        const double t = -(double)(16 * blk + lj - W / 2) / (double)U - (double)m;
                        k[((((size_t)which * 2 + c) * blocks + blk) * 64 + lane) * 16 + i] =
                            (float)class_kernel(c, P, t);
                    }
    return k;
}

// The same tables in float64 for the 64 tile's float64 refine (spx_kernels.h RefineF64).  Stage 1 as above; stage 2
// follows v_mfma_f64_16x16x4_f64's result layout (register r of lane group lk holds row lk + 4 r):
//   [1][c][blk][lane][4 t + r] = K_c( -(16 blk + lj - W/2)/U - (16 t + lk + 4 r - 32) )
inline std::vector<double> make_ktab_f64(int P, int U, int W) {
    const int blocks = W / 16;
    std::vector<double> k((size_t)2 * 2 * blocks * 64 * 16);
    for (int which = 0; which < 2; ++which)
        for (int c = 0; c < 2; ++c)
            for (int blk = 0; blk < blocks; ++blk)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 16; ++i) {
                        const int lk = lane >> 4, lj = lane & 15;
                        const int m = which == 0 ? (4 * i + lk - 32)
                                                 : (16 * (i >> 2) + lk + 4 * (i & 3) - 32);
                        const double t = -(double)(16 * blk + lj - W / 2) / (double)U - (double)m;
                        k[((((size_t)which * 2 + c) * blocks + blk) * 64 + lane) * 16 + i] = class_kernel(c, P, t);
                    }
    return k;
}

// 32 tile (period 64, 32-lag class planes), spx_kernels32.h fine_window32:
//   [0][c][blk][lane][s]       = K_c( -(16 blk + lj - W/2)/U - (4 s + lk - 16) ),        s in [0,8)
//   [1][c][blk][lane][4 t + r] = K_c( -(16 blk + lj - W/2)/U - (16 t + 4 lk + r - 16) ), t in [0,2)
inline std::vector<float> make_ktab32(int U, int W) {
    const int blocks = W / 16;
    std::vector<float> k((size_t)2 * 2 * blocks * 64 * 8);
    for (int which = 0; which < 2; ++which)
        for (int c = 0; c < 2; ++c)
            for (int blk = 0; blk < blocks; ++blk)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 8; ++i) {
                        const int lk = lane >> 4, lj = lane & 15;
                        const int m = which == 0 ? (4 * i + lk - 16)
                                                 : (16 * (i >> 2) + 4 * lk + (i & 3) - 16);
                        const double t = -(double)(16 * blk + lj - W / 2) / (double)U - (double)m;
                        k[((((size_t)which * 2 + c) * blocks + blk) * 64 + lane) * 8 + i] =
                            (float)class_kernel(c, 64, t);
                    }
    return k;
}

// Period-256 real interpolation kernel (128 tile) and its lane-major MFMA tables
// (spx_kernels128.h fine_window128), P = 192 or 256:
//   [0][blk][lane][s]       = K(-(16 blk + lj - W/2)/U - (4 s + lk - P/2)),        s in [0,P/4)
//   [1][blk][lane][4 T + r] = K(-(16 blk + lj - W/2)/U - (CW w + 4 TPW lk + TPW r + t - P/2)),
//                             CW = P/4 columns per wave, TPW = CW/16 column tiles, T = TPW w + t
inline double kernel_big(int P, double t) {
    double s = 1.0 + std::cos(kPi * t);
    for (int k = 1; k < P / 2; ++k) s += 2.0 * std::cos(2.0 * kPi * k * t / (double)P);
    return s / (double)P;
}
// P = 192 (96 tile) or 256 (128 tile); P/4 entries per (part, block, lane)
inline std::vector<float> make_ktab_big(int P, int U, int W) {
    const int blocks = W / 16, n = P / 4;
    std::vector<float> k((size_t)2 * blocks * 64 * n);
    for (int which = 0; which < 2; ++which)
        for (int blk = 0; blk < blocks; ++blk)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < n; ++i) {
                    const int lk = lane >> 4, lj = lane & 15;
                    // [1]: i = 4 T + r, T = TPW w + t  ->  column CW w + 4 TPW lk + TPW r + t
                    const int cw = P / 4, tpw = cw / 16;
                    const int T = i >> 2, r = i & 3, w = T / tpw, tt = T % tpw;
                    const int m = which == 0 ? (4 * i + lk - P / 2)
                                             : (cw * w + 4 * tpw * lk + tpw * r + tt - P / 2);
                    const double t = -(double)(16 * blk + lj - W / 2) / (double)U - (double)m;
                    k[(((size_t)which * blocks + blk) * 64 + lane) * n + i] = (float)kernel_big(P, t);
                }
    return k;
}
inline std::vector<float> make_ktab256(int U, int W) { return make_ktab_big(256, U, W); }

// The same tables in float64 for the float64-accumulating refine stage of the paths above 85 px
// (fine_window128 / fine_window_big use v_mfma_f64_16x16x4_f64).  Part [1] follows that
// instruction's C/D map (row = lk + 4 r instead of 4 lk + r):
//   [1][blk][lane][4 T + r] = K(-(16 blk + lj - W/2)/U - (CW w + TPW (lk + 4 r) + t - P/2))
inline std::vector<double> make_ktab_big_f64(int P, int U, int W) {
    const int blocks = W / 16, n = P / 4;
    std::vector<double> k((size_t)2 * blocks * 64 * n);
    for (int which = 0; which < 2; ++which)
        for (int blk = 0; blk < blocks; ++blk)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < n; ++i) {
                    const int lk = lane >> 4, lj = lane & 15;
                    const int cw = P / 4, tpw = cw / 16;
                    const int T = i >> 2, r = i & 3, w = T / tpw, tt = T % tpw;
                    const int m = which == 0 ? (4 * i + lk - P / 2)
                                             : (cw * w + tpw * (lk + 4 * r) + tt - P / 2);
                    const double t = -(double)(16 * blk + lj - W / 2) / (double)U - (double)m;
                    k[(((size_t)which * blocks + blk) * 64 + lane) * n + i] = kernel_big(P, t);
                }
    return k;
}

}  // namespace host
}  // namespace spx
