// spx_kernels_big.h -- the general path: cutouts of 129..682 pixels per side.
//
// The reference builds its cutouts from segment bounding boxes + padding (cutout.py:159-175) with
// no upper bound on the size.  Everything above 128 px takes this path: FFT period P = 64 C with
// the class count C per axis a RUN-TIME value (4..16: the smallest multiple of 64 that leaves the
// 'same' window of cc.py:114-126 free of circular aliasing, P > 2n - 2 - (n-1)/2), the same
// decomposition as spx_kernels128.h:
//   Z[C k'+c] = FFT64{ fold_c(z)[x'] w_P^(c x') },  fold_c(z)[x'] = sum_s w_C^(c s) z[x' + 64 s]
// per axis, C*C classes of 64x64 complex points, each transformed by one wave in registers with
// the machinery of spx_kernels.h.  It is written for generality, not speed: the cutout is staged
// one 64x64 block at a time (each of a round's four waves folds the block into its own class),
// the class results g_c go to a per-workgroup workspace, and the radix-C combine is done in two
// separable passes through that workspace:
//   h[cy][sx][l'] = sum_cx conj(w_C)^(cx sx) g[cy][cx][l']
//   conv[l' + 64 s] = Im( sum_cy conj(w_C)^(cy sy) h[cy][sx][l'] ) / (2 P^2 bal)
// followed by the arg-max over the flipped 'same' window, the MFMA refine (period-P Dirichlet
// kernel, one 16-column tile at a time) and the 5x5 fit, exactly as on the other paths.
//
// Workspace per workgroup (floats): g planes 2 C^2 64^2 | h planes 2 C^2 64^2 | conv P (P+4).
#pragma once

namespace spx {

constexpr int kBigMaxC = 16;                     // P = 1024: cutouts up to 682 px

struct BigGeom {
    int C, P, CS;
    size_t plane, g_off, h_off, conv_off, ws_floats;
    SPX_DEVICE BigGeom(int c) : C(c), P(64 * c), CS(64 * c + 4), plane(64 * 64) {
        g_off = 0;
        h_off = (size_t)2 * c * c * plane;
        conv_off = 2 * h_off;
        ws_floats = conv_off + (size_t)P * CS;
    }
    SPX_DEVICE int wrap(int x) const {           // x mod P for any int
        x %= P;
        return x < 0 ? x + P : x;
    }
};
// host and device: class count for an (ny, nx) cutout and the workspace it needs
inline int big_class_count(int ny, int nx) {
    const int n = ny > nx ? ny : nx;
    const int need = 2 * n - 2 - (n - 1) / 2 + 1;
    return (need + 63) / 64;
}
inline size_t big_ws_floats(int C) {
    return (size_t)4 * C * C * 64 * 64 + (size_t)(64 * C) * (64 * C + 4);
}

struct LdsGen {
    static constexpr int ZS = 72, XS = 68;
    static constexpr int TW_OFF = 0;                              // cf[P], P <= 1024
    static constexpr int SCR_OFF = TW_OFF + 64 * kBigMaxC * 8;    // 1 KiB scratch
    static constexpr int R_OFF = SCR_OFF + 1024;
    static constexpr int XCH_WAVE_BYTES = 64 * XS * 4;
    static constexpr int XCH_BYTES = 4 * XCH_WAVE_BYTES;
    static constexpr int FB_OFF = R_OFF;                          // fine windows reuse the exchange region
    static constexpr int total(int W) {
        return R_OFF + (XCH_BYTES > 4 * W * W * 4 ? XCH_BYTES : 4 * W * W * 4);
    }
};

// one 64x64 block (sy, sx) of z = ref + i*flip(img), normalised, into the LDS staging planes;
// returns this thread's share of sum ref^2 / sum img^2 in ssq
template <typename TIn, bool NARROW, bool NX4 = false>
SPX_DEVICE void stage_block_big_rows(unsigned char* lds, const TIn* __restrict__ ref,
                                const TIn* __restrict__ img, int ny, int nx, int sy, int sx,
                                const NormStatsT<TIn>& ns, float (&ssq)[2]) {
    typedef LdsGen L;
    const int tid = fresh_tid();
    float* zre = reinterpret_cast<float*>(lds + L::R_OFF);
    float* zim = zre + 64 * L::ZS;
    ChunkLoad<TIn> ld[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * kThreads;
        ld[i] = chunk_issue<TIn, NARROW, NX4>(ref, img, ny, nx, (idx >> 4) + 64 * sy, ((idx & 15) << 2) + 64 * sx);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + i * kThreads;
        const int yl = idx >> 4, x = (idx & 15) << 2;
        float rr[4], mm[4];
        chunk_unpack(ld[i], ns, rr, mm);
        *reinterpret_cast<f32x4*>(zre + yl * L::ZS + x) = f32x4{rr[0], rr[1], rr[2], rr[3]};
        *reinterpret_cast<f32x4*>(zim + yl * L::ZS + x) = f32x4{mm[0], mm[1], mm[2], mm[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) { ssq[0] += rr[e] * rr[e]; ssq[1] += mm[e] * mm[e]; }
    }
}

template <typename TIn>
SPX_DEVICE void stage_block_big(unsigned char* lds, const TIn* __restrict__ ref,
                                const TIn* __restrict__ img, int ny, int nx, int sy, int sx,
                                const NormStatsT<TIn>& ns, float (&ssq)[2]) {
    // cutouts narrower than a load chunk: element loads (chunk_issue), uniform per item
    if (nx < 4) stage_block_big_rows<TIn, true>(lds, ref, img, ny, nx, sy, sx, ns, ssq);
    else if ((nx & 3) == 0) stage_block_big_rows<TIn, false, true>(lds, ref, img, ny, nx, sy, sx, ns, ssq);
    else stage_block_big_rows<TIn, false>(lds, ref, img, ny, nx, sy, sx, ns, ssq);
}

// sums of squares over the whole cutout (as staged) -> balance factor; one pass over the pair
template <typename TIn>
SPX_DEVICE float balance_big(unsigned char* lds, const TIn* __restrict__ ref,
                             const TIn* __restrict__ img, int ny, int nx, const NormStatsT<TIn>& ns) {
    const int tid = fresh_tid();
    float ssq[2] = {0.0f, 0.0f};
    const int chunks = (nx + 3) >> 2;
    for (int g = tid; g < ny * chunks; g += kThreads) {
        const int y = g / chunks, x = (g - y * chunks) << 2;
        const ChunkLoad<TIn> c = nx < 4 ? chunk_issue<TIn, true>(ref, img, ny, nx, y, x)
                                        : chunk_issue<TIn, false>(ref, img, ny, nx, y, x);
        float rr[4], mm[4];
        chunk_unpack(c, ns, rr, mm);
#pragma unroll
        for (int e = 0; e < 4; ++e) { ssq[0] += rr[e] * rr[e]; ssq[1] += mm[e] * mm[e]; }
    }
    return balance_factor(lds + LdsGen::SCR_OFF, ssq);
}

// One round: the four classes c = 4 round + wave (inactive beyond C*C) -> complex planes g_c
template <typename TIn>
SPX_DEVICE void class_round_big(unsigned char* lds, const BigGeom& G, const TIn* __restrict__ ref,
                                const TIn* __restrict__ img, int ny, int nx,
                                const NormStatsT<TIn>& ns, float bal, int round,
                                float* __restrict__ ws) {
    typedef LdsGen L;
    const int C = G.C;
    const int tid = fresh_tid();
    const int wave = tid >> 6, lane = tid & 63;
    const int cls = 4 * round + wave;
    const bool active = cls < C * C;
    const int cy = active ? cls / C : 0, cx = active ? cls % C : 0;
    const int l1 = lane >> 3, l0 = lane & 7;
    const cf* tw = reinterpret_cast<const cf*>(lds + L::TW_OFF);
    const float* zre = reinterpret_cast<const float*>(lds + L::R_OFF);
    const float* zim = zre + 64 * L::ZS;
    float* xch = reinterpret_cast<float*>(lds + L::R_OFF + wave * L::XCH_WAVE_BYTES);

    cf v[8][8];
#pragma unroll
    for (int r = 0; r < 64; ++r) v[r >> 3][r & 7] = cf{0.0f, 0.0f};
    const int by = (ny + 63) >> 6, bx = (nx + 63) >> 6;
    for (int s = 0; s < by * bx; ++s) {
        const int sy = s / bx, sx = s - sy * bx;
        rt::block_sync_lds();          // the previous block (or round) is no longer being read
        float dummy[2] = {0.0f, 0.0f};
        stage_block_big(lds, ref, img, ny, nx, sy, sx, ns, dummy);
        rt::block_sync_lds();
        // d += w_C^(cy sy + cx sx) (re, bal im)
        const cf w = tw[64 * ((cy * sy + cx * sx) % C)];
#pragma unroll
        for (int y1 = 0; y1 < 8; ++y1)
#pragma unroll
            for (int x1 = 0; x1 < 8; ++x1) {
                const int a = (l1 + 8 * y1) * L::ZS + l0 + 8 * x1;
                rt::cmac_ip(v[y1][x1], cf{zre[a], bal * zim[a]}, w);
            }
    }
    rt::block_sync_lds();              // staging area is about to become the exchange buffers
    // class pre-twiddle w_P^{c (8 y1)}
#pragma unroll
    for (int y1 = 1; y1 < 8; ++y1) {
        const cf w = tw[8 * cy * y1];
#pragma unroll
        for (int x1 = 0; x1 < 8; ++x1) rt::cmul_ip(v[y1][x1], w);
    }
#pragma unroll
    for (int x1 = 1; x1 < 8; ++x1) {
        const cf w = tw[8 * cx * x1];
#pragma unroll
        for (int y1 = 0; y1 < 8; ++y1) rt::cmul_ip(v[y1][x1], w);
    }
    fft8_y<1>(v);
    fft8_x<1>(v);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        const cf wy = tw[l1 * (cy + C * kb)];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[kb][j] = cmul(v[kb][j], wy);
    }
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        const cf wx = tw[l0 * (cx + C * kb)];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j][kb] = cmul(v[j][kb], wx);
    }
    transpose_tile_cplx<L::XS>(v, xch, lane);
    fft8_y<1>(v);
    fft8_x<1>(v);
#pragma unroll
    for (int r = 0; r < 64; ++r) v[r >> 3][r & 7] = cmul(v[r >> 3][r & 7], v[r >> 3][r & 7]);
    fft8_y<-1>(v);
    fft8_x<-1>(v);
#pragma unroll
    for (int y0 = 1; y0 < 8; ++y0) {
        const cf wy = tw[y0 * (cy + C * l1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[y0][j] = cmulc(v[y0][j], wy);
    }
#pragma unroll
    for (int x0 = 1; x0 < 8; ++x0) {
        const cf wx = tw[x0 * (cx + C * l0)];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j][x0] = cmulc(v[j][x0], wx);
    }
    transpose_tile_cplx<L::XS>(v, xch, lane);
    fft8_y<-1>(v);
    fft8_x<-1>(v);
    // class post-twiddle conj(w_P^{8 (cy y1 + cx x1)})
#pragma unroll
    for (int y1 = 0; y1 < 8; ++y1) {
        const cf wy = tw[8 * cy * y1];
#pragma unroll
        for (int x1 = 0; x1 < 8; ++x1) {
            const cf wx = tw[8 * cx * x1];
            v[y1][x1] = cmulc(cmulc(v[y1][x1], wy), wx);
        }
    }
    float* g = ws + G.g_off + (size_t)(cls * 2) * G.plane;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int y1 = 0; y1 < 8; ++y1)
#pragma unroll
            for (int x1 = 0; x1 < 8; ++x1)
                xch[plane_elem(l1 + 8 * y1, l0 + 8 * x1)] = part ? v[y1][x1].y : v[y1][x1].x;
        rt::wave_sync();
        if (active) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                reinterpret_cast<f32x4*>(g + part * G.plane)[i * 64 + lane] =
                    reinterpret_cast<const f32x4*>(xch)[plane_slot(i * 64 + lane)];
        }
        rt::wave_sync();
    }
}

// separable radix-C combine through the workspace (see the header)
SPX_DEVICE void combine_big(unsigned char* lds, const BigGeom& G, float* __restrict__ ws, float out_scale) {
    const int C = G.C;
    const int tid = fresh_tid();
    const cf* tw = reinterpret_cast<const cf*>(lds + LdsGen::TW_OFF);
    const float* g = ws + G.g_off;
    float* h = ws + G.h_off;
    float* conv = ws + G.conv_off;
    // pass X: h[cy][sx] = sum_cx conj(w_C)^(cx sx) g[cy][cx]
    for (int item = tid; item < C * C * 1024; item += kThreads) {
        const int i4 = item & 1023, cs = item >> 10;
        const int cy = cs / C, sx = cs - cy * C;
        f32x4 are = f32x4{0.f, 0.f, 0.f, 0.f}, aim = are;
        for (int cx = 0; cx < C; ++cx) {
            const cf w = tw[64 * ((cx * sx) % C)];               // w_C^(cx sx); conj applied below
            const f32x4 gre = reinterpret_cast<const f32x4*>(g + (size_t)((cy * C + cx) * 2) * G.plane)[i4];
            const f32x4 gim = reinterpret_cast<const f32x4*>(g + (size_t)((cy * C + cx) * 2 + 1) * G.plane)[i4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {                         // (gre + i gim)(w.x - i w.y)
                are[e] += gre[e] * w.x + gim[e] * w.y;
                aim[e] += gim[e] * w.x - gre[e] * w.y;
            }
        }
        reinterpret_cast<f32x4*>(h + (size_t)((cy * C + sx) * 2) * G.plane)[i4] = are;
        reinterpret_cast<f32x4*>(h + (size_t)((cy * C + sx) * 2 + 1) * G.plane)[i4] = aim;
    }
    rt::block_sync();
    // pass Y: conv[l' + 64 s] = out_scale Im sum_cy conj(w_C)^(cy sy) h[cy][sx]
    for (int item = tid; item < C * C * 1024; item += kThreads) {
        const int i4 = item & 1023, ss = item >> 10;
        const int sy = ss / C, sx = ss - sy * C;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int cy = 0; cy < C; ++cy) {
            const cf w = tw[64 * ((cy * sy) % C)];
            const f32x4 hre = reinterpret_cast<const f32x4*>(h + (size_t)((cy * C + sx) * 2) * G.plane)[i4];
            const f32x4 him = reinterpret_cast<const f32x4*>(h + (size_t)((cy * C + sx) * 2 + 1) * G.plane)[i4];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += him[e] * w.x - hre[e] * w.y;    // Im((hre + i him) conj(w))
        }
        const int ly = i4 >> 4, lx = (i4 & 15) << 2;
        float* row = conv + (size_t)(ly + 64 * sy) * G.CS + lx + 64 * sx;
        *reinterpret_cast<f32x4*>(row) = acc * out_scale;
    }
    rt::block_sync();
}

SPX_DEVICE float window_value_big(const float* __restrict__ conv, const BigGeom& G, int ny, int nx,
                                  int qy, int qx) {
    return conv[(size_t)conv_index(ny, qy) * G.CS + conv_index(nx, qx)];
}

// arg-max over the flipped 'same' window (index qy * nx + qx); MODE 1 writes the window into the
// interlaced image instead and accumulates the arg-max over that (NaN ranked as +inf)
template <int MODE>
SPX_DEVICE void window_scan_big(const float* __restrict__ conv, const BigGeom& G, int ny, int nx,
                                float* __restrict__ icc, int ox, int oy, float& bv, int& bi) {
    const int tid = fresh_tid();
    for (int idx = tid; idx < ny * nx; idx += kThreads) {
        const int qy = idx / nx, qx = idx - qy * nx;
        const float val = window_value_big(conv, G, ny, nx, qy, qx);
        if constexpr (MODE == 0) {
            if (better(val, idx, bv, bi)) { bv = val; bi = idx; }
        } else {
            const int gi = (2 * qy + oy) * (2 * nx) + 2 * qx + ox;
            icc[gi] = val;
            if (better(nan_as_inf(val), gi, bv, bi)) { bv = nan_as_inf(val); bi = gi; }
        }
    }
}

// Fine window by MFMA (float64 accumulation, see fine_window128), period P at run time: wave w
// takes the window columns [CW w - P/2, CW w - P/2 + CW), CW = P/4, one 16-column tile at a time
// (tables: make_ktab_big_f64).
template <int WB>
SPX_DEVICE void fine_window_big(unsigned char* lds, const BigGeom& G, const double* __restrict__ ktab,
                                const float* __restrict__ conv, int ny, int nx, int qyc, int qxc) {
    typedef rt::f64x4 f64x4;
    constexpr int W = 16 * WB;
    const int P = G.P, NQ = P / 16, CW = P / 4, TPW = CW / 16;
    const int tid = fresh_tid();
    const int wave = tid >> 6, lane = tid & 63;
    const int lk = lane >> 4, lj = lane & 15;
    float* fbuf = reinterpret_cast<float*>(lds + LdsGen::FB_OFF) + wave * W * W;
    qyc = qyc < 0 ? 0 : (qyc > ny ? ny : qyc);
    qxc = qxc < 0 ? 0 : (qxc > nx ? nx : qxc);
    const int lyc = conv_index(ny, qyc), lxc = conv_index(nx, qxc);
    const double* kty = ktab + (size_t)lane * NQ * 4;
    const double* ktx = ktab + ((size_t)WB * 64 + lane) * NQ * 4;
    f64x4 f[WB][WB];
#pragma unroll
    for (int bb = 0; bb < WB; ++bb)
#pragma unroll
        for (int ab = 0; ab < WB; ++ab) f[bb][ab] = f64x4{0., 0., 0., 0.};
    for (int t = 0; t < TPW; ++t) {
        const int col = G.wrap(lxc + CW * wave + TPW * lj + t - P / 2);
        f64x4 acc[WB];
#pragma unroll
        for (int ab = 0; ab < WB; ++ab) acc[ab] = f64x4{0., 0., 0., 0.};
        for (int s4 = 0; s4 < NQ; ++s4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = G.wrap(lyc + 4 * (4 * s4 + e) + lk - P / 2);
                const double a = (double)conv[(size_t)row * G.CS + col];
#pragma unroll
                for (int ab = 0; ab < WB; ++ab)
                    acc[ab] = rt::mfma_f64_16x16x4(a, kty[((size_t)ab * 64 * NQ + s4) * 4 + e], acc[ab]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int bb = 0; bb < WB; ++bb) {
                const double ka = ktx[((size_t)bb * 64 * NQ + TPW * wave + t) * 4 + r];
#pragma unroll
                for (int ab = 0; ab < WB; ++ab)
                    f[bb][ab] = rt::mfma_f64_16x16x4(ka, acc[ab][r], f[bb][ab]);
            }
    }
#pragma unroll
    for (int bb = 0; bb < WB; ++bb)
#pragma unroll
        for (int ab = 0; ab < WB; ++ab)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                fbuf[(bb * 16 + lk + 4 * r) * W + ab * 16 + lj] = (float)f[bb][ab][r];
    rt::block_sync_lds();
}

template <int W> SPX_DEVICE float fine_value_big(const unsigned char* lds, int b, int a) {
    const float* fbuf = reinterpret_cast<const float*>(lds + LdsGen::FB_OFF);
    float acc = fbuf[b * W + a];
#pragma unroll
    for (int w = 1; w < 4; ++w) acc += fbuf[w * W * W + b * W + a];
    return acc;
}

// cutout pair -> full PxP convolution in the workspace (ends with a full barrier)
template <typename TIn>
SPX_DEVICE void conv_full_big(unsigned char* lds, const BigGeom& G, const TIn* __restrict__ ref,
                              const TIn* __restrict__ img, int ny, int nx,
                              const NormStatsT<TIn>& ns, float* __restrict__ ws) {
    const float bal = balance_big(lds, ref, img, ny, nx, ns);
    const int rounds = (G.C * G.C + 3) / 4;
    for (int r = 0; r < rounds; ++r) class_round_big(lds, G, ref, img, ny, nx, ns, bal, r, ws);
    rt::block_sync();
    const float out_scale = 0.5f / ((float)G.P * (float)G.P * bal);
    combine_big(lds, G, ws, out_scale);
}

SPX_DEVICE void load_twiddles_big(unsigned char* lds, const cf* __restrict__ tw_g, int P) {
    cf* tw = reinterpret_cast<cf*>(lds + LdsGen::TW_OFF);
    for (int i = rt::thread_id(); i < P; i += kThreads) tw[i] = tw_g[i];
    rt::block_sync_lds();
}

// pair mode, general path
template <int WB, typename TIn>
SPX_TKERNEL(256) void pair_big_kernel(const TIn* __restrict__ ref, const TIn* __restrict__ img,
                                      int64_t nbatch, int ny, int nx, int U, int cc_type, int C,
                                      const cf* __restrict__ tw_g, const double* __restrict__ ktab,
                                      double* __restrict__ out, int* __restrict__ status,
                                      float* __restrict__ workspace) {
    typedef LdsGen L;
    SPX_DYN_LDS(lds);
    const BigGeom G(C);
    load_twiddles_big(lds, tw_g, G.P);
    float* ws = workspace + (size_t)rt::block_id() * G.ws_floats;
    const float* conv = ws + G.conv_off;
    unsigned char* scr = lds + L::SCR_OFF;
    const int64_t stride = (int64_t)ny * nx;
    for (int64_t p = first_item(rt::block_id(), rt::grid_size()); p < nbatch; p += rt::grid_size()) {
        const TIn* r = ref + p * stride;
        const TIn* m = img + p * stride;
        const int tid = fresh_tid();
        const NormStatsT<TIn> ns = norm_stats(scr, r, m, 1, 0, ny, nx, cc_type);
        conv_full_big(lds, G, r, m, ny, nx, ns, ws);
        float bv = -__builtin_inff();
        int bi = kNoIndex;
        window_scan_big<0>(conv, G, ny, nx, nullptr, 0, 0, bv, bi);
        block_argmax(scr, bv, bi, 0);
        const bool nonfinite = bi == kNoIndex;       // NaN everywhere (see pair_body in spx_kernels.h)
        if (nonfinite) bi = 0;
        int qyc = bi / nx, qxc = bi - (bi / nx) * nx;
        PeakResult pk;
        if (nonfinite) {
            pk.x = 0.0; pk.y = 0.0; pk.status = ST_NONFINITE;
        } else if constexpr (WB == 0) {
            pk = peak_fit_wave0(scr, qxc, qyc, nx, ny, [&](int x, int y) {
                return window_value_big(conv, G, ny, nx, y, x);
            });
        } else {
            constexpr int W = 16 * (WB > 0 ? WB : 1);
            const int NX = U * nx, NY = U * ny;
            int imax = 0, jmax = 0;
            bool inside = false;
            for (int iter = 0; iter < 4; ++iter) {
                fine_window_big<(WB > 0 ? WB : 1)>(lds, G, ktab, conv, ny, nx, qyc, qxc);
                const int fx0 = U * qxc - W / 2, fy0 = U * qyc - W / 2;
                float fv = -__builtin_inff();
                int fi = kNoIndex;
                for (int idx = tid; idx < W * W; idx += kThreads) {
                    const int a = idx / W, b = idx % W;
                    const int gy = fy0 + a, gx = fx0 + b;
                    if (gy >= 0 && gy < NY && gx >= 0 && gx < NX) {
                        const float val = fine_value_big<W>(lds, b, a);
                        if (better(val, idx, fv, fi)) { fv = val; fi = idx; }
                    }
                }
                block_argmax(scr, fv, fi, 1);
                if (fi == kNoIndex) { imax = jmax = -1; break; }
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
                rt::block_sync_lds();
            }
            if (inside) {
                const int fx0 = U * qxc - W / 2, fy0 = U * qyc - W / 2;
                pk = peak_fit_wave0(scr, imax, jmax, NX, NY, [&](int x, int y) {
                    return fine_value_big<W>(lds, x - fx0, y - fy0);
                });
            } else if (imax < 0) {
                pk.x = 0.0; pk.y = 0.0; pk.status = ST_NONFINITE;
            } else {
                pk.x = (double)imax; pk.y = (double)jmax; pk.status = ST_WINDOW;
            }
        }
        if (tid == 0) {
            out[2 * p] = pk.x / (double)U - (double)((nx - 1) / 2);
            out[2 * p + 1] = pk.y / (double)U - (double)((ny - 1) / 2);
            if (status) status[p] = pk.status;
        }
        rt::block_sync();
    }
}

// reference (5-image) mode, general path
template <typename TIn>
SPX_TKERNEL(256) void disp5_big_kernel(const TIn* __restrict__ ref, const TIn* __restrict__ im4,
                                       int64_t nbatch, int ny, int nx, int cc_type, int C,
                                       const cf* __restrict__ tw_g, float* __restrict__ icc_all,
                                       double* __restrict__ out_all, int* __restrict__ status,
                                       float* __restrict__ workspace) {
    typedef LdsGen L;
    SPX_DYN_LDS(lds);
    const BigGeom G(C);
    load_twiddles_big(lds, tw_g, G.P);
    float* ws = workspace + (size_t)rt::block_id() * G.ws_floats;
    const float* conv = ws + G.conv_off;
    unsigned char* scr = lds + L::SCR_OFF;
    const int64_t stride = (int64_t)ny * nx;
    const int NX = 2 * nx, NY = 2 * ny;
    for (int64_t p = first_item(rt::block_id(), rt::grid_size()); p < nbatch; p += rt::grid_size()) {
        const TIn* r = ref + p * stride;
        const TIn* m4 = im4 + 4 * p * stride;
        float* icc = icc_all + 4 * p * stride;
        const int tid = fresh_tid();
        const NormStatsT<TIn> ns = norm_stats(scr, r, m4, 4, stride, ny, nx, cc_type);
        float bv = -__builtin_inff();
        int bi = kNoIndex;
        for (int q = 0; q < 4; ++q) {
            conv_full_big(lds, G, r, m4 + q * stride, ny, nx, ns, ws);
            window_scan_big<1>(conv, G, ny, nx, icc, q & 1, q >> 1, bv, bi);
            rt::block_sync();
        }
        block_argmax(scr, bv, bi, 0);
        const bool nonfinite = !(bv < __builtin_inff());        // see disp5_body (spx_kernels.h)
        const int jmax = bi / NX, imax = bi % NX;
        PeakResult pk;
        if (nonfinite) {
            pk.x = (double)imax; pk.y = (double)jmax; pk.status = ST_NONFINITE;
        } else {
            pk = peak_fit_wave0(scr, imax, jmax, NX, NY, [&](int x, int y) { return icc[(size_t)y * NX + x]; });
        }
        if (tid == 0) {
            out_all[2 * p] = 0.5 * pk.x - (double)((NX - 1) / 4);
            out_all[2 * p + 1] = 0.5 * pk.y - (double)((NY - 1) / 4);
            if (status) status[p] = pk.status;
        }
        rt::block_sync();
    }
}

}  // namespace spx
