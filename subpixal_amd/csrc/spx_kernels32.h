// spx_kernels32.h -- the 32x32 tile (cutouts up to 32 pixels per side): FFT period P = 64
// (scipy's next_fast_len(2*32-1)), ONE WAVE PER PAIR, four pairs per 256-thread workgroup and
// no workgroup barrier anywhere in the per-pair path.
//
// Same algorithm as spx_kernels.h.  The padded 64x64 spectrum splits into 4 parity classes
// Z[2k'+c] = FFT32{ z[x] w_64^(c x) }[k'], each a 32x32 complex FFT (32 = 8 x 4).  A wave keeps all
// four classes in registers:
//   round A   lane = (class, y0, x0) with y0, x0 in [0,4); registers (y1, x1) in 8x8  -> radix-8 in y and x
//   transposition (the same lane <-> register swap through the wave's LDS buffer)
//   round B   lane = (kyb, kxb) in 8x8; registers (class, y0, x0) -> radix-4 in y0 and x0, per class
// so x = x0 + 4 x1, k' = kb + 8 ka (ka in [0,4)).  W = Z^2, the inverse runs backwards and leaves the
// four real 32x32 class planes in the wave's buffer;  conv[l] = P^-2 sum_c (-1)^(c.[l>=32]) d_c[l mod 32].
// Arg-max, MFMA refine (all four classes accumulate into one window in registers) and the 5x5 fit are
// done by the same wave.
#pragma once

namespace spx {

struct Lds32 {
    static constexpr int P = 64;
    static constexpr int ZS = 36;             // staged-input row stride (floats): conflict-free tile reads
    static constexpr int XS = 68;
    static constexpr int TW_OFF = 0;                        // cf[64]
    static constexpr int SCR_OFF = TW_OFF + P * 8;          // 4 x 256 B per-wave scratch
    static constexpr int R_OFF = SCR_OFF + 1024;
    static constexpr int PLANES_BYTES = 4 * 32 * 32 * 4;    // 16 KiB, at the head of the wave's buffer
    // per-wave buffer: transposition (64 x 68 floats) | staging | planes + fine window
    static constexpr int wave_bytes(int W) {
        return (PLANES_BYTES + W * W * 4) > 64 * XS * 4 ? (PLANES_BYTES + W * W * 4) : 64 * XS * 4;
    }
    static constexpr int total(int W) { return R_OFF + 4 * wave_bytes(W); }
};

SPX_DEVICE float wave_sum_f(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += rt::shfl_xor(v, m);
    return v;
}

// element (class c, y0, x0) of the round-B register tile
#define SPX_E32(c, y0, x0) v[(((c) * 16 + (y0) * 4 + (x0)) >> 3)][(((c) * 16 + (y0) * 4 + (x0)) & 7)]

template <int DIR> SPX_DEVICE void fft4_inplace(cf& a0, cf& a1, cf& a2, cf& a3) {
    cf y0, y1, y2, y3;
    fft4<DIR, false>(a0, a1, a2, a3, y0, y1, y2, y3);
    a0 = y0; a1 = y1; a2 = y2; a3 = y3;
}
// radix-4 along y0 and x0 of every class sub-tile
template <int DIR> SPX_DEVICE void fft4_classes(cf (&v)[8][8]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int x0 = 0; x0 < 4; ++x0)
            fft4_inplace<DIR>(SPX_E32(c, 0, x0), SPX_E32(c, 1, x0), SPX_E32(c, 2, x0), SPX_E32(c, 3, x0));
#pragma unroll
        for (int y0 = 0; y0 < 4; ++y0)
            fft4_inplace<DIR>(SPX_E32(c, y0, 0), SPX_E32(c, y0, 1), SPX_E32(c, y0, 2), SPX_E32(c, y0, 3));
    }
}

// cc.py:131-156 statistics for one wave's pair (npool images), wave-level reductions
template <typename TIn>
SPX_DEVICE NormStatsT<TIn> norm_stats_wave(const TIn* __restrict__ ref, const TIn* __restrict__ ims,
                                     int npool, int64_t im_stride, int npx, int cc_type) {
    NormStatsT<TIn> ns;
    ns.active = 0;
    ns.im_mean = 0; ns.im_rstd = 1; ns.ref_mean = 0; ns.ref_rstd = 1;
    if (cc_type == CC_PLAIN) return ns;
    const int lane = fresh_tid() & 63;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int i = lane; i < npx; i += 64) {
        bool any = false;
        for (int q = 0; q < npool; ++q) {
            const TIn m = ims[q * im_stride + i];
            if (m != (TIn)0) { a0 += 1.0; a1 += (double)m; any = true; }
        }
        if (any) { a2 += 1.0; a3 += (double)ref[i]; }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
    const double im_mean = a1 / a0, ref_mean = a3 / a2;
    double b0 = 0.0, b1 = 0.0;
    for (int i = lane; i < npx; i += 64) {
        bool any = false;
        for (int q = 0; q < npool; ++q) {
            const TIn m = ims[q * im_stride + i];
            if (m != (TIn)0) { const double d = (double)m - im_mean; b0 += d * d; any = true; }
        }
        if (any) { const double d = (double)ref[i] - ref_mean; b1 += d * d; }
    }
    b0 = wave_sum(b0); b1 = wave_sum(b1);
    ns.active = 1;
    const bool zero = (cc_type == CC_ZNCC);
    ns.im_mean = zero ? (TIn)im_mean : (TIn)0;
    ns.im_rstd = (TIn)(1.0 / sqrt(b0 / a0));
    ns.ref_mean = zero ? (TIn)ref_mean : (TIn)0;
    ns.ref_rstd = (TIn)(1.0 / sqrt(b1 / a2));
    return ns;
}

// (ref, flipped img) of one pair -> the four real class planes in the wave's buffer `wbuf`.
// Returns the exact power-of-two balance factor applied to the image.
// CPLX: the register-saving transposition (transpose_tile_cplx), used by the reference-mode kernel
template <typename TIn, bool CPLX = false>
SPX_DEVICE float cc_planes32(const cf* tw, float* wbuf, const TIn* __restrict__ ref,
                             const TIn* __restrict__ img, int ny, int nx, const NormStatsT<TIn>& ns) {
    typedef Lds32 L;
    const int lane = fresh_tid() & 63;
    const int cls = lane >> 4, cy = cls >> 1, cx = cls & 1;       // round-A lane = (class, y0, x0)
    const int ay0 = (lane >> 2) & 3, ax0 = lane & 3;
    const int l1 = lane >> 3, l0 = lane & 7;                        // round-B lane = (kyb, kxb)
    float* zre = wbuf;
    float* zim = wbuf + 32 * L::ZS;

    // ---- stage (this wave only), with sums of squares for the balance factor
    float sr = 0.0f, sm = 0.0f;
    for (int idx = lane; idx < 32 * 32; idx += 64) {
        const int y = idx >> 5, x = idx & 31;
        float r = 0.0f, m = 0.0f;
        if (y < ny && x < nx) {
            const TIn ri = ref[y * nx + x];
            const TIn mi = img[(ny - 1 - y) * nx + (nx - 1 - x)];    // flipped: cc.py:114
            r = (float)ri;
            m = (float)mi;
            if (ns.active) {
                m = norm_im(mi, ns);
                r = norm_ref(ri, ns);
            }
        }
        zre[y * L::ZS + x] = r;
        zim[y * L::ZS + x] = m;
        sr += r * r;
        sm += m * m;
    }
    sr = wave_sum_f(sr);
    sm = wave_sum_f(sm);
    const float bal = balance_from_ssq(sr, sm);
    rt::wave_sync();

    // ---- forward round A: x = x0 + 4 x1
    cf v[8][8];
#pragma unroll
    for (int y1 = 0; y1 < 8; ++y1)
#pragma unroll
        for (int x1 = 0; x1 < 8; ++x1) {
            const int a = (ay0 + 4 * y1) * L::ZS + ax0 + 4 * x1;
            v[y1][x1] = cf{zre[a], bal * zim[a]};
        }
    rt::wave_sync();                        // staging area is reused by the transposition
    // class pre-twiddle w_64^{c (4 y1)} (the class sits in the lane: no uniform skip)
#pragma unroll
    for (int y1 = 1; y1 < 8; ++y1) {
        const cf w = tw[4 * cy * y1];
#pragma unroll
        for (int x1 = 0; x1 < 8; ++x1) v[y1][x1] = cmul(v[y1][x1], w);
    }
#pragma unroll
    for (int x1 = 1; x1 < 8; ++x1) {
        const cf w = tw[4 * cx * x1];
#pragma unroll
        for (int y1 = 0; y1 < 8; ++y1) v[y1][x1] = cmul(v[y1][x1], w);
    }
    fft8_y<1>(v);
    fft8_x<1>(v);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        const cf wy = tw[ay0 * (cy + 2 * kb)];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[kb][j] = cmul(v[kb][j], wy);
    }
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        const cf wx = tw[ax0 * (cx + 2 * kb)];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j][kb] = cmul(v[j][kb], wx);
    }
    if constexpr (CPLX) transpose_tile_cplx<L::XS>(v, wbuf, lane); else transpose_tile<L::XS>(v, wbuf, lane);
    // ---- forward round B (radix-4 per class), W = Z^2, inverse round A'
    fft4_classes<1>(v);
#pragma unroll
    for (int r = 0; r < 64; ++r) v[r >> 3][r & 7] = cmul(v[r >> 3][r & 7], v[r >> 3][r & 7]);
    fft4_classes<-1>(v);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int y0 = 0; y0 < 4; ++y0) {
            const cf wy = tw[y0 * ((c >> 1) + 2 * l1)];
#pragma unroll
            for (int x0 = 0; x0 < 4; ++x0) {
                const cf wx = tw[x0 * ((c & 1) + 2 * l0)];
                cf e = SPX_E32(c, y0, x0);
                if (y0) e = cmulc(e, wy);         // index 0: w^0
                if (x0) e = cmulc(e, wx);
                SPX_E32(c, y0, x0) = e;
            }
        }
    if constexpr (CPLX) transpose_tile_cplx<L::XS>(v, wbuf, lane); else transpose_tile<L::XS>(v, wbuf, lane);
    // ---- inverse round B'
    fft8_y<-1>(v);
    fft8_x<-1>(v);
    // class post-twiddle conj(w_64^{4 (cy y1 + cx x1)}); keep half the imaginary part
    float* plane = wbuf + cls * (32 * 32);
#pragma unroll
    for (int y1 = 0; y1 < 8; ++y1) {
        const cf wy = tw[4 * cy * y1];
#pragma unroll
        for (int x1 = 0; x1 < 8; ++x1) {
            const cf wx = tw[4 * cx * x1];
            const cf a = cmulc(v[y1][x1], wy);
            plane[(ay0 + 4 * y1) * 32 + ax0 + 4 * x1] = 0.5f * (a.y * wx.x - a.x * wx.y);
        }
    }
    rt::wave_sync();
    return bal;
}

SPX_DEVICE float window_value32(const float* wbuf, int ny, int nx, int qy, int qx, float out_scale) {
    const int ly = conv_index(ny, qy), lx = conv_index(nx, qx);
    const int my = ly & 31, mx = lx & 31;
    const int sy = (ly >> 5) & 1, sx = (lx >> 5) & 1;
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float d = wbuf[c * 1024 + my * 32 + mx];
        const int neg = ((c >> 1) & sy) ^ ((c & 1) & sx);
        acc += neg ? -d : d;
    }
    return acc * out_scale;
}

// wave-wide coarse arg-max over the flipped 'same' window, planes walked in storage order
SPX_DEVICE void coarse_argmax32(const float* wbuf, int ny, int nx, float out_scale, float& bv, int& bi) {
    const int lane = fresh_tid() & 63;
    const int loy = (ny - 1) / 2, lox = (nx - 1) / 2;
    bv = -__builtin_inff();
    bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = lane + i * 64;               // 256 float4 chunks per plane
        const int my = g >> 3, mx4 = (g & 7) << 2;
        const int ly = my + (my < loy ? 32 : 0);
        const int qy = (ny - 1) + loy - ly;
        const int sy = ly >> 5;
        f32x4 d[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = *reinterpret_cast<const f32x4*>(wbuf + c * 1024 + my * 32 + mx4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int mx = mx4 + e;
            const int lx = mx + (mx < lox ? 32 : 0);
            const int qx = (nx - 1) + lox - lx;
            const int sx = lx >> 5;
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int neg = ((c >> 1) & sy) ^ ((c & 1) & sx);
                acc += neg ? -d[c][e] : d[c][e];
            }
            const float val = acc * out_scale;
            const int idx = qy * nx + qx;
            if (qy >= 0 && qx >= 0 && better(val, idx, bv, bi)) { bv = val; bi = idx; }
        }
    }
    wave_argmax(bv, bi);
}

// Fine window by MFMA, all four classes accumulated in registers by this wave.  Tables
// (make_ktab32), lane = 16 lk + lj, parity c in {0,1}:
//   [0][c][blk][lane][s]       = K_c(-(16 blk + lj - W/2)/U - (4 s + lk - 16)),        s in [0,8)
//   [1][c][blk][lane][4 t + r] = K_c(-(16 blk + lj - W/2)/U - (16 t + 4 lk + r - 16)), t in [0,2)
template <int WB>
SPX_DEVICE void fine_window32(float* wbuf, const float* __restrict__ ktab, int ny, int nx, int qyc,
                              int qxc) {
    constexpr int W = 16 * WB;
    const int lane = fresh_tid() & 63;
    const int lk = lane >> 4, lj = lane & 15;
    float* fbuf = wbuf + 4 * 1024;
    const int lyc = conv_index(ny, qyc), lxc = conv_index(nx, qxc);
    ktab = rt::launder(ktab);
    f32x4 f[WB][WB];
#pragma unroll
    for (int bb = 0; bb < WB; ++bb)
#pragma unroll
        for (int ab = 0; ab < WB; ++ab) f[bb][ab] = f32x4{0.f, 0.f, 0.f, 0.f};
    int col[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) col[t] = (lxc + 16 * t + lj - 16) & 31;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int cy = c >> 1, cx = c & 1;
        const float* plane = wbuf + c * 1024;
        const f32x4* kty = reinterpret_cast<const f32x4*>(ktab) + ((size_t)(0 * 2 + cy) * WB * 64 + lane) * 2;
        const f32x4* ktx = reinterpret_cast<const f32x4*>(ktab) + ((size_t)(1 * 2 + cx) * WB * 64 + lane) * 2;
        f32x4 acc[WB][2];
#pragma unroll
        for (int ab = 0; ab < WB; ++ab)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[ab][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 2; ++s4) {
            f32x4 kb[WB];
#pragma unroll
            for (int ab = 0; ab < WB; ++ab) kb[ab] = kty[ab * 64 * 2 + s4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = lyc + 4 * (4 * s4 + e) + lk - 16;
                const int row = m & 31;
                const float sgn = (cy && ((m >> 5) & 1)) ? -1.0f : 1.0f;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float afrag = plane[row * 32 + col[t]];
#pragma unroll
                    for (int ab = 0; ab < WB; ++ab)
                        acc[ab][t] = rt::mfma_16x16x4(afrag, sgn * kb[ab][e], acc[ab][t]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 ka[WB];
#pragma unroll
            for (int bb = 0; bb < WB; ++bb) ka[bb] = ktx[bb * 64 * 2 + t];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = lxc + 16 * t + 4 * lk + r - 16;
                const float sgn = (cx && ((m >> 5) & 1)) ? -1.0f : 1.0f;
#pragma unroll
                for (int bb = 0; bb < WB; ++bb)
#pragma unroll
                    for (int ab = 0; ab < WB; ++ab)
                        f[bb][ab] = rt::mfma_16x16x4(sgn * ka[bb][r], acc[ab][t][r], f[bb][ab]);
            }
        }
    }
    const float scale = 1.0f / (float)(Lds32::P * Lds32::P);
#pragma unroll
    for (int bb = 0; bb < WB; ++bb)
#pragma unroll
        for (int ab = 0; ab < WB; ++ab)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                fbuf[(bb * 16 + 4 * lk + r) * W + ab * 16 + lj] = f[bb][ab][r] * scale;
    rt::wave_sync();
}

// 5x5 fit by one wave (peak_fit_wave0 without the wave test)
template <typename ValFn>
SPX_DEVICE PeakResult peak_fit_wave(double* fit, int imax, int jmax, int NX, int NY, ValFn val) {
    const int lane = fresh_tid() & 63;
    PeakResult r;
    r.x = (double)imax; r.y = (double)jmax; r.status = ST_EDGE;
    if (imax == 0 || jmax == 0) return r;
    int x1 = imax - 2, y1 = jmax - 2;
    if (x1 > NX - 5) x1 = NX - 5;
    if (y1 > NY - 5) y1 = NY - 5;
    if (x1 < 0) x1 = 0;
    if (y1 < 0) y1 = 0;
    const int k = lane < 25 ? lane : 24;
    const float v = val(x1 + k % 5, y1 + k / 5);
    return quad_fit_wave(v, lane, x1, y1, imax, jmax, NX, NY);
}

template <int WB, typename TIn>
SPX_DEVICE void pair32_wave(const cf* tw, float* wbuf, double* fit, const TIn* __restrict__ ref,
                            const TIn* __restrict__ img, int ny, int nx, int U, int cc_type,
                            const float* __restrict__ ktab, double* __restrict__ out,
                            int* __restrict__ status, const TIn* __restrict__ next_ref,
                            const TIn* __restrict__ next_img, float& warm) {
    ny = rt::launder_uniform(ny);
    nx = rt::launder_uniform(nx);
    U = rt::launder_uniform(U);
    const int lane = fresh_tid() & 63;
    const NormStatsT<TIn> ns = norm_stats_wave(ref, img, 1, 0, ny * nx, cc_type);
    rt::consume(warm);        // the warm-up load of this pair has landed (or was never issued)
    const float bal = cc_planes32(tw, wbuf, ref, img, ny, nx, ns);
    // L2 warm-up of this wave's NEXT pair while the current one is in its arg-max / refine / fit
    // tail (as in the 64 tile, spx_kernels.h: warm_next_pair): one element per 128-byte line, lanes
    // 0..31 the reference, 32..63 the image; the value is only kept alive until the next staging
    if (next_ref) {
        const int off = (lane & 31) * (int)(128 / sizeof(TIn));
        const TIn* q = (lane < 32 ? next_ref : next_img) + (off < ny * nx ? off : 0);
        warm = (float)*q;
    }
    const float oscale = 1.0f / ((float)(Lds32::P * Lds32::P) * bal);
    float bv;
    int bi;
    coarse_argmax32(wbuf, ny, nx, oscale, bv, bi);
    const bool nonfinite = bi == kNoIndex;       // NaN everywhere (see pair_body in spx_kernels.h)
    if (nonfinite) bi = 0;
    int qyc = bi / nx, qxc = bi - (bi / nx) * nx;
    PeakResult pk;
    if (nonfinite) {
        pk.x = 0.0; pk.y = 0.0; pk.status = ST_NONFINITE;
    } else if constexpr (WB == 0) {
        pk = peak_fit_wave(fit, qxc, qyc, nx, ny, [&](int x, int y) {
            return window_value32(wbuf, ny, nx, y, x, oscale);
        });
    } else {
        constexpr int W = 16 * (WB > 0 ? WB : 1);
        const int NX = U * nx, NY = U * ny;
        const float* fbuf = wbuf + 4 * 1024;
        int imax = 0, jmax = 0;
        bool inside = false;
        for (int iter = 0; iter < 4; ++iter) {
            fine_window32<(WB > 0 ? WB : 1)>(wbuf, ktab, ny, nx, qyc, qxc);
            const int fx0 = U * qxc - W / 2, fy0 = U * qyc - W / 2;
            float fv = -__builtin_inff();
            int fi = 0x7fffffff;
            for (int idx = lane; idx < W * W; idx += 64) {
                const int a = idx / W, b = idx % W;
                const int gy = fy0 + a, gx = fx0 + b;
                if (gy >= 0 && gy < NY && gx >= 0 && gx < NX) {
                    const float val = fbuf[b * W + a];
                    if (better(val, idx, fv, fi)) { fv = val; fi = idx; }
                }
            }
            wave_argmax(fv, fi);
            if (fi == kNoIndex) { imax = jmax = -1; break; }      // non-finite window (overflow)
            const int a = fi / W, b = fi % W;
            jmax = fy0 + a;
            imax = fx0 + b;
            int x1 = imax - 2, y1 = jmax - 2;
            if (x1 > NX - 5) x1 = NX - 5;
            if (y1 > NY - 5) y1 = NY - 5;
            if (x1 < 0) x1 = 0;
            if (y1 < 0) y1 = 0;
            const bool okx = (x1 >= fx0 && x1 + 4 < fx0 + W) || imax == 0;
            const bool oky = (y1 >= fy0 && y1 + 4 < fy0 + W) || jmax == 0;
            if (okx && oky) { inside = true; break; }
            if (!okx) qxc += (b < W / 2) ? -1 : 1;
            if (!oky) qyc += (a < W / 2) ? -1 : 1;
            qxc = qxc < 0 ? 0 : (qxc > nx ? nx : qxc);
            qyc = qyc < 0 ? 0 : (qyc > ny ? ny : qyc);
            rt::wave_sync();
        }
        if (inside) {
            const int fx0 = U * qxc - W / 2, fy0 = U * qyc - W / 2;
            pk = peak_fit_wave(fit, imax, jmax, NX, NY, [&](int x, int y) {
                return fbuf[(x - fx0) * W + (y - fy0)];
            });
        } else if (imax < 0) {
            pk.x = 0.0; pk.y = 0.0; pk.status = ST_NONFINITE;
        } else {
            pk.x = (double)imax; pk.y = (double)jmax; pk.status = ST_WINDOW;
        }
    }
    if (lane == 0) {
        out[0] = pk.x / (double)U - (double)((nx - 1) / 2);
        out[1] = pk.y / (double)U - (double)((ny - 1) / 2);
        if (status) status[0] = pk.status;
    }
    rt::wave_sync();
}

SPX_DEVICE void load_twiddles32(unsigned char* lds, const cf* __restrict__ tw_g) {
    cf* tw = reinterpret_cast<cf*>(lds + Lds32::TW_OFF);
    for (int i = rt::thread_id(); i < Lds32::P; i += kThreads) tw[i] = tw_g[i];
    rt::block_sync_lds();
}

template <int WB, typename TIn = float>
SPX_TKERNEL(256) void pair32_kernel(const TIn* __restrict__ ref, const TIn* __restrict__ img,
                                    int64_t nbatch, int ny, int nx, int U, int cc_type,
                                    const cf* __restrict__ tw_g, const float* __restrict__ ktab,
                                    double* __restrict__ out, int* __restrict__ status) {
    typedef Lds32 L;
    SPX_DYN_LDS(lds);
    load_twiddles32(lds, tw_g);
    // the wave index as a SCALAR: the pair pointers derived from it (this pair's and the next one's) then
    // live in scalar registers instead of occupying -- and spilling -- vector ones
    const int wave = rt::read_lane(rt::thread_id() >> 6, 0);
    const cf* tw = reinterpret_cast<const cf*>(lds + L::TW_OFF);
    float* wbuf = reinterpret_cast<float*>(lds + L::R_OFF + wave * L::wave_bytes(16 * (WB > 0 ? WB : 1)));
    double* fit = reinterpret_cast<double*>(lds + L::SCR_OFF + wave * 256);
    const int64_t stride = (int64_t)ny * nx;
    // one pair per wave; waves of a workgroup never synchronise with each other
    const int64_t step = rt::grid_size() * 4;
    float warm = 0.0f;
    for (int64_t p = first_item(rt::block_id(), rt::grid_size()) * 4 + wave; p < nbatch; p += step) {
        const bool more = p + step < nbatch;
        pair32_wave<WB, TIn>(tw, wbuf, fit, ref + p * stride, img + p * stride, ny, nx, U, cc_type, ktab,
                        out + 2 * p, status ? status + p : nullptr,
                        more ? ref + (p + step) * stride : nullptr, more ? img + (p + step) * stride : nullptr, warm);
    }
}

// reference (5-image) mode on the 32 tile: one wave per source
template <typename TIn = float>
SPX_TKERNEL(256) void disp5_32_kernel(const TIn* __restrict__ ref, const TIn* __restrict__ im4,
                                      int64_t nbatch, int ny, int nx, int cc_type,
                                      const cf* __restrict__ tw_g, float* __restrict__ icc_all,
                                      double* __restrict__ out_all, int* __restrict__ status,
                                      ItemTable items) {
    typedef Lds32 L;
    SPX_DYN_LDS(lds);
    load_twiddles32(lds, tw_g);
    // the wave index as a SCALAR: the per-source pointers and statistics derived from it then live in
    // scalar registers across the four dithers instead of occupying (and spilling) vector ones
    const int wave = rt::read_lane(rt::thread_id() >> 6, 0);
    const cf* tw = reinterpret_cast<const cf*>(lds + L::TW_OFF);
    float* wbuf = reinterpret_cast<float*>(lds + L::R_OFF + wave * L::wave_bytes(16));
    double* fit = reinterpret_cast<double*>(lds + L::SCR_OFF + wave * 256);
    const int ny_u = ny, nx_u = nx;
    for (int64_t p = first_item(rt::block_id(), rt::grid_size()) * 4 + wave; p < nbatch; p += rt::grid_size() * 4) {
        const int lane = fresh_tid() & 63;
        const ItemView it = item_view(items, p, ny_u, nx_u);       // per-item shape (see spx_kernels.h)
        if (!it.ok) { if (!it.skip) item_refused(out_all, status, p, lane == 0); continue; }
        ny = it.ny;
        nx = it.nx;
        const int64_t stride = (int64_t)ny * nx;
        const int NX = 2 * nx, NY = 2 * ny;
        const TIn* r = ref + it.off;
        const TIn* m4 = im4 + 4 * it.off;
        float* icc = icc_all + 4 * it.off;
        NormStatsT<TIn> ns = norm_stats_wave(r, m4, 4, stride, ny * nx, cc_type);
        ns.im_mean = rt::read_lane(ns.im_mean, 0);          // wave-uniform values -> scalar registers
        ns.im_rstd = rt::read_lane(ns.im_rstd, 0);
        ns.ref_mean = rt::read_lane(ns.ref_mean, 0);
        ns.ref_rstd = rt::read_lane(ns.ref_rstd, 0);
        float bv = -__builtin_inff();
        int bi = 0x7fffffff;
        for (int q = 0; q < 4; ++q) {
            const int ox = q & 1, oy = q >> 1;
            const float bal = cc_planes32<TIn, true>(tw, wbuf, r, m4 + q * stride, ny, nx, ns);
            const float oscale = 1.0f / ((float)(L::P * L::P) * bal);
            for (int idx = lane; idx < ny * nx; idx += 64) {
                const int qy = idx / nx, qx = idx - qy * nx;
                const float val = window_value32(wbuf, ny, nx, qy, qx, oscale);
                const int gi = (2 * qy + oy) * NX + 2 * qx + ox;
                icc[gi] = val;
                if (better(nan_as_inf(val), gi, bv, bi)) { bv = nan_as_inf(val); bi = gi; }
            }
            rt::wave_sync_mem();     // icc (global) is read back by other lanes for the fit
        }
        wave_argmax(bv, bi);
        const bool nonfinite = !(bv < __builtin_inff());        // see disp5_body (spx_kernels.h)
        const int jmax = bi / NX, imax = bi % NX;
        PeakResult pk;
        if (nonfinite) {
            pk.x = (double)imax; pk.y = (double)jmax; pk.status = ST_NONFINITE;
        } else {
            pk = peak_fit_wave(fit, imax, jmax, NX, NY, [&](int x, int y) { return icc[(size_t)y * NX + x]; });
        }
        if (lane == 0) {
            out_all[2 * p] = 0.5 * pk.x - (double)((NX - 1) / 4);
            out_all[2 * p + 1] = 0.5 * pk.y - (double)((NY - 1) / 4);
            if (status) status[p] = pk.status;
        }
        rt::wave_sync();
    }
}

}  // namespace spx
