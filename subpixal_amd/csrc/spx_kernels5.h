// spx_kernels5.h -- reference mode (cc.find_displacement, cc.py:21-95) on the 64 tile with FIVE transforms per
// source instead of eight (round 3).
//
// The reference correlates one cutout against four dithers (cc.py:114-117: four fftconvolve calls, i.e. 8 forward +
// 4 inverse real transforms; SURVEY 8 a-2 notes that 5 + 4 suffice).  disp5_kernel of spx_kernels.h ran the
// pair kernel's trick four times -- z = ref + i flip(im_q), Im IFFT(Z^2): one forward and one inverse complex
// transform per dither, and the reference cutout re-staged and re-transformed with each.  Here
//     R  = FFT(ref)                                             once per source,
//     Zq = FFT(flip(im_a) + i flip(im_b))                        once per dither PAIR (a, b) = (00, 10), (01, 11),
//     IFFT(R Zq) = ref * flip(im_a) + i ref * flip(im_b)         (* = convolution; both parts are real correlations)
// -- five complex transforms, the reference read once, no squaring (so no balance factor), and the two
// correlations of a pair come out as the real and imaginary part of the SAME complex numbers: exactly the two
// horizontally interlaced samples icc[2 qy + oy][2 qx + {0, 1}] (cc.py:121-126), stored as one 8-byte pair
// instead of two 4-byte stores a dither apart.
//
// The class spectrum R (64x64 complex per wave = 128 registers) has to stay resident next to the transform in
// flight (another 128): the kernel is built for ONE wave per SIMD (up to 512 registers), one workgroup per CU,
// and uses the CU's whole LDS share for a second set of class planes (the imaginary parts).  A wave alone on
// its SIMD issues packed FP32 at full rate; what it gives up is the other workgroup's cover for LDS and
// barrier latency -- measured against disp5_kernel in profiles/r03.
//
// Class transforms, twiddles, plane layout and the per-source tail (arg-max over the interlaced image, 5x5 fit,
// centroid.py:114-236) are those of spx_kernels.h.  Needs spx_kernels.h first.
#pragma once

namespace spx {
namespace p5 {

template <bool FOLD> struct L5 {
    typedef Lds<2> L;
    typedef StageGeom<2, FOLD> G;
    static constexpr int PLANES2_OFF = L::R_OFF + L::XCH_BYTES;          // imaginary-part planes, 4 x 16 KiB
    static constexpr int PLANE2_BYTES = 64 * 64 * 4;
    static constexpr int TOTAL = PLANES2_OFF + 4 * PLANE2_BYTES;        // 134 KiB: one workgroup per CU
};

// ---------------------------------------------------------------------------
// Staging.  MODE 0: the reference cutout alone (real plane = ref, normalised as cc.py:153-154; the imaginary
// plane is not written and not read).  MODE 1: a dither pair, BOTH flipped (cc.py:114: im[::-1, ::-1]) and
// normalised as cc.py:144-148: real plane = flip(a), imaginary plane = flip(b).
// ---------------------------------------------------------------------------
// The loads of one staging step (issue5) and their normalisation + LDS writes (commit5) are separate calls, so
// that a step's loads can be in flight under the previous step's transform: with one wave per SIMD nothing
// else covers a trip to memory.
template <bool FOLD, int MODE, typename TIn> struct Stage5Loads {
    typedef StageGeom<2, FOLD> G;
    static constexpr int kIters = (G::ROWS * G::CHUNKS + kThreads - 1) / kThreads;
    ChunkLoad<TIn> la[kIters], lb[MODE ? kIters : 1];
};
template <bool FOLD, int MODE, typename TIn, bool NARROW, bool NX4>
SPX_DEVICE void issue5_rows(Stage5Loads<FOLD, MODE, TIn>& ld, const TIn* __restrict__ a, const TIn* __restrict__ b,
                            int ny, int nx) {
    typedef StageGeom<2, FOLD> G;
    const int tid = fresh_tid();
    // chunk_issue(ref, img, ...) fetches ref[y][x..] (field r) and the flipped img (field t): MODE 0 uses r of
    // (a, a), MODE 1 uses t of (a, a) and t of (b, b); the unused halves are never loaded (dead code)
#pragma unroll
    for (int i = 0; i < Stage5Loads<FOLD, MODE, TIn>::kIters; ++i) {
        const int idx = tid + i * kThreads;
        const int y = FOLD ? idx / G::CHUNKS : idx >> 4;
        const int x = (FOLD ? idx - y * G::CHUNKS : (idx & 15)) << 2;
        ld.la[i] = chunk_issue<TIn, NARROW, NX4>(a, a, ny, nx, y, x);
        if (MODE) ld.lb[i] = chunk_issue<TIn, NARROW, NX4>(b, b, ny, nx, y, x);
    }
}
template <bool FOLD, int MODE, typename TIn>
SPX_DEVICE void issue5(Stage5Loads<FOLD, MODE, TIn>& ld, const TIn* __restrict__ a, const TIn* __restrict__ b,
                       int ny, int nx) {
    if (nx < 4) issue5_rows<FOLD, MODE, TIn, true, false>(ld, a, b, ny, nx);              // (uniform per item)
    else if ((nx & 3) == 0) issue5_rows<FOLD, MODE, TIn, false, true>(ld, a, b, ny, nx);
    else issue5_rows<FOLD, MODE, TIn, false, false>(ld, a, b, ny, nx);
}
template <bool FOLD, int MODE, typename TIn>
SPX_DEVICE void commit5(unsigned char* lds, const Stage5Loads<FOLD, MODE, TIn>& ld, const NormStatsT<TIn>& ns) {
    typedef Lds<2> L;
    typedef StageGeom<2, FOLD> G;
    const int tid = fresh_tid();
    float* zre = reinterpret_cast<float*>(lds + L::R_OFF);
    float* zim = zre + G::ROWS * G::ZS;
#pragma unroll
    for (int i = 0; i < Stage5Loads<FOLD, MODE, TIn>::kIters; ++i) {
        const int idx = tid + i * kThreads;
        if (FOLD && idx >= G::ROWS * G::CHUNKS) break;
        const int y = FOLD ? idx / G::CHUNKS : idx >> 4;
        const int x = (FOLD ? idx - y * G::CHUNKS : (idx & 15)) << 2;
        float ra[4], fa[4], rb[4], fb[4];
        chunk_unpack(ld.la[i], ns, ra, fa);               // ra: as a reference cutout, fa: flipped, as an image
        if (MODE) chunk_unpack(ld.lb[i], ns, rb, fb);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = FOLD ? y * G::ZS + x + e : y * G::ZS + G::perm(x >> 2, e);
            zre[o] = MODE ? fa[e] : ra[e];
            if (MODE) zim[o] = fb[e];
        }
    }
}

// ---------------------------------------------------------------------------
// Forward class transform of the staged planes: v = Z[cy + 2 l1 + 16 kya][cx + 2 l0 + 16 kxa] on lane
// (l1, l0), register (kya, kxa) -- cc_planes of spx_kernels.h up to its spectral point.  REAL: the imaginary
// plane is absent (the reference cutout).  Contains the barrier after the tile load.
// ---------------------------------------------------------------------------
template <bool FOLD, bool REAL>
SPX_DEVICE void forward5(unsigned char* lds, cf (&v)[8][8]) {
    typedef Lds<2> L;
    typedef StageGeom<2, FOLD> G;
    const int tid = fresh_tid();
    const int lane = tid & 63, wave = tid >> 6;
    const int cy = wave >> 1, cx = wave & 1;
    const int l1 = lane >> 3, l0 = lane & 7;
    const cf* tw = reinterpret_cast<const cf*>(lds + L::TW_OFF);
    const float* zre = reinterpret_cast<const float*>(lds + L::R_OFF);
    const float* zim = zre + G::ROWS * G::ZS;
    float* xch = reinterpret_cast<float*>(lds + L::R_OFF + wave * L::XCH_WAVE_BYTES);
    if constexpr (!FOLD) {
#pragma unroll
        for (int y1 = 0; y1 < 8; ++y1) {
            const f32x4* pr = reinterpret_cast<const f32x4*>(zre + (l1 + 8 * y1) * G::ZS + l0 * G::PS12);
            const f32x4* pi = reinterpret_cast<const f32x4*>(zim + (l1 + 8 * y1) * G::ZS + l0 * G::PS12);
            const f32x4 r0 = pr[0], r1 = pr[1];
            f32x4 i0 = f32x4{0.f, 0.f, 0.f, 0.f}, i1 = i0;
            if (!REAL) { i0 = pi[0]; i1 = pi[1]; }
#pragma unroll
            for (int x1 = 0; x1 < 4; ++x1) {
                v[y1][x1] = cf{r0[x1], i0[x1]};
                v[y1][x1 + 4] = cf{r1[x1], i1[x1]};
            }
        }
    } else {
        // the class's radix-2 fold of the samples beyond index 63 (cc_planes)
        const float fsx = cx ? -1.0f : 1.0f, fsy = cy ? -1.0f : 1.0f;
#pragma unroll
        for (int y1 = 0; y1 < 8; ++y1)
#pragma unroll
            for (int x1 = 0; x1 < 8; ++x1) {
                const int a = (l1 + 8 * y1) * G::ZS + l0 + 8 * x1;
                float re = zre[a], im = REAL ? 0.0f : zim[a];
                if (x1 < 3) {
                    re = __builtin_fmaf(fsx, zre[a + 64], re);
                    if (!REAL) im = __builtin_fmaf(fsx, zim[a + 64], im);
                }
                if (y1 < 3) {
                    float re2 = zre[a + 64 * G::ZS], im2 = REAL ? 0.0f : zim[a + 64 * G::ZS];
                    if (x1 < 3) {
                        re2 = __builtin_fmaf(fsx, zre[a + 64 * G::ZS + 64], re2);
                        if (!REAL) im2 = __builtin_fmaf(fsx, zim[a + 64 * G::ZS + 64], im2);
                    }
                    re = __builtin_fmaf(fsy, re2, re);
                    if (!REAL) im = __builtin_fmaf(fsy, im2, im);
                }
                v[y1][x1] = cf{re, im};
            }
    }
    rt::block_sync_lds();                       // all waves have read the staged input
    if (cy) {
#pragma unroll
        for (int y1 = 1; y1 < 8; ++y1) {
            const cf w = tw[8 * cy * y1];
#pragma unroll
            for (int x1 = 0; x1 < 8; ++x1) rt::cmul_ip(v[y1][x1], w);
        }
    }
    if (cx) {
#pragma unroll
        for (int x1 = 1; x1 < 8; ++x1) {
            const cf w = tw[8 * cx * x1];
#pragma unroll
            for (int y1 = 0; y1 < 8; ++y1) rt::cmul_ip(v[y1][x1], w);
        }
    }
    fft8_y<1>(v);
    fft8_x<1>(v);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        const cf wy = tw[l1 * (cy + 2 * kb)];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[kb][j] = cmul(v[kb][j], wy);
    }
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        const cf wx = tw[l0 * (cx + 2 * kb)];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j][kb] = cmul(v[j][kb], wx);
    }
    transpose_tile<L::XS>(v, xch, lane);
    fft8_y<1>(v);
    fft8_x<1>(v);
}

// Inverse class transform of v (spectral layout of forward5) and the two plane sets: real parts (the first
// dither of the pair) in the exchange buffers' heads (pair_kernel's plane layout), imaginary parts (the
// second) behind the exchange region.  Ends with a barrier.
template <bool FOLD>
SPX_DEVICE void inverse5(unsigned char* lds, cf (&v)[8][8]) {
    typedef Lds<2> L;
    const int tid = fresh_tid();
    const int lane = tid & 63, wave = tid >> 6;
    const int cy = wave >> 1, cx = wave & 1;
    const int l1 = lane >> 3, l0 = lane & 7;
    const cf* tw = reinterpret_cast<const cf*>(lds + L::TW_OFF);
    float* xch = reinterpret_cast<float*>(lds + L::R_OFF + wave * L::XCH_WAVE_BYTES);
    fft8_y<-1>(v);
    fft8_x<-1>(v);
#pragma unroll
    for (int y0 = 1; y0 < 8; ++y0) {
        const cf wy = tw[y0 * (cy + 2 * l1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[y0][j] = cmulc(v[y0][j], wy);
    }
#pragma unroll
    for (int x0 = 1; x0 < 8; ++x0) {
        const cf wx = tw[x0 * (cx + 2 * l0)];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j][x0] = cmulc(v[j][x0], wx);
    }
    transpose_tile<L::XS>(v, xch, lane);
    fft8_y<-1>(v);
    fft8_x<-1>(v);
    // class post-twiddle conj(w_P^{8 (cy y1 + cx x1)}); both parts are kept
    float* pre = reinterpret_cast<float*>(lds + L::R_OFF + wave * L::PLANE_STRIDE_BYTES);
    float* pim = reinterpret_cast<float*>(lds + L5<FOLD>::PLANES2_OFF + wave * L5<FOLD>::PLANE2_BYTES);
    if (cy) {
#pragma unroll
        for (int y1 = 1; y1 < 8; ++y1) {
            const cf wy = tw[8 * cy * y1];
#pragma unroll
            for (int x1 = 0; x1 < 8; ++x1) rt::cmulc_ip(v[y1][x1], wy);
        }
    }
#pragma unroll
    for (int x1 = 0; x1 < 8; ++x1) {
        const cf wx = cx ? tw[8 * cx * x1] : cf{1.0f, 0.0f};
#pragma unroll
        for (int y1 = 0; y1 < 8; ++y1) {
            const cf a = cx && x1 ? cmulc(v[y1][x1], wx) : v[y1][x1];
            const int row = l1 + 8 * y1;
            const int o = row * L::PS + plane_col(row, l0 + 8 * x1);
            pre[o] = a.x;
            pim[o] = a.y;
        }
    }
    rt::block_sync_lds();
}

// The flipped 'same' windows of BOTH correlations of a dither pair -> their interlaced positions
// icc[2 qy + oy][2 qx + {0, 1}] (cc.py:121-126), one 8-byte store per window element, and the running arg-max
// over the interlaced image (interlace_window of spx_kernels.h for two plane sets).
template <bool FOLD>
SPX_DEVICE void interlace_pair(const unsigned char* lds, int ny, int nx, float out_scale,
                               float* __restrict__ icc, int oy, float& bv, int& bi) {
    typedef Lds<2> L;
    const int tid = fresh_tid();
    const int loy = (ny - 1) / 2, lox = (nx - 1) / 2;
    const int NX = 2 * nx;
    const int mx4 = (tid & 15) << 2;
    float m = bv;
    int best = bi;
    auto put = [&](int row, int qx, float va, float vb) {
        if (row >= 0 && qx >= 0) {
            const int g = row + 2 * qx;
            *reinterpret_cast<PackedF2*>(icc + g) = PackedF2{{va, vb}};
            va = nan_as_inf(va);
            vb = nan_as_inf(vb);
            if (better(va, g, m, best)) { m = va; best = g; }
            if (better(vb, g + 1, m, best)) { m = vb; best = g + 1; }
        }
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int my = (tid >> 4) + 16 * i;
        f32x4 dr[4], di[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int o = (my * L::PS + plane_col(my, mx4)) * 4;
            dr[c] = *reinterpret_cast<const f32x4*>(lds + L::R_OFF + c * L::PLANE_STRIDE_BYTES + o);
            di[c] = *reinterpret_cast<const f32x4*>(lds + L5<FOLD>::PLANES2_OFF + c * L5<FOLD>::PLANE2_BYTES + o);
        }
        if constexpr (!FOLD) {
            const bool wy = my < loy;
            const int qy = (ny - 1) + loy - my - (wy ? 64 : 0);
            const float fy = wy ? -1.0f : 1.0f;
            const int row = qy >= 0 ? (2 * qy + oy) * NX : -1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int mx = mx4 + e;
                const bool wx = mx < lox;
                const int qx = (nx - 1) + lox - mx - (wx ? 64 : 0);
                const float fx = wx ? -1.0f : 1.0f;
                // same association as interlace_window: (d00 + fx d01) + fy (d10 + fx d11)
                const float ua = __builtin_fmaf(fx, dr[1][e], dr[0][e]), ta = __builtin_fmaf(fx, dr[3][e], dr[2][e]);
                const float ub = __builtin_fmaf(fx, di[1][e], di[0][e]), tb = __builtin_fmaf(fx, di[3][e], di[2][e]);
                put(row, qx, __builtin_fmaf(fy, ta, ua) * out_scale, __builtin_fmaf(fy, tb, ub) * out_scale);
            }
        } else {
            // 65..85 px: a plane element stands for up to two convolution indices per axis (coarse_argmax_fold)
            const int qyb = (ny - 1) + loy - my - 64;
            const int rowa = my >= loy ? (2 * ((ny - 1) + loy - my) + oy) * NX : -1;
            const int rowb = qyb >= 0 ? (2 * qyb + oy) * NX : -1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int mx = mx4 + e;
                const int qxa = mx >= lox ? (nx - 1) + lox - mx : -1;
                const int qxb = (nx - 1) + lox - mx - 64;
                const float upa = dr[0][e] + dr[1][e], uma = dr[0][e] - dr[1][e];
                const float tpa = dr[2][e] + dr[3][e], tma = dr[2][e] - dr[3][e];
                const float upb = di[0][e] + di[1][e], umb = di[0][e] - di[1][e];
                const float tpb = di[2][e] + di[3][e], tmb = di[2][e] - di[3][e];
                put(rowa, qxa, (upa + tpa) * out_scale, (upb + tpb) * out_scale);
                put(rowa, qxb, (uma + tma) * out_scale, (umb + tmb) * out_scale);
                put(rowb, qxa, (upa - tpa) * out_scale, (upb - tpb) * out_scale);
                put(rowb, qxb, (uma - tma) * out_scale, (umb - tmb) * out_scale);
            }
        }
    }
    bv = m;
    bi = best;
}

template <bool FOLD, typename TIn>
SPX_DEVICE void disp5p_body(const TIn* __restrict__ ref, const TIn* __restrict__ im4, int ny, int nx,
                            int cc_type, float* __restrict__ icc, double* __restrict__ out,
                            int* __restrict__ status, unsigned char* lds) {
    typedef Lds<2> L;
    unsigned char* scr = lds + L::SCR_OFF;
    const int64_t stride = (int64_t)ny * nx;
    NormStatsT<TIn> ns = norm_stats(scr, ref, im4, 4, stride, ny, nx, cc_type);
    ns.im_mean = rt::read_lane(ns.im_mean, 0);          // workgroup-uniform: scalar registers
    ns.im_rstd = rt::read_lane(ns.im_rstd, 0);
    ns.ref_mean = rt::read_lane(ns.ref_mean, 0);
    ns.ref_rstd = rt::read_lane(ns.ref_rstd, 0);
    ns.active = rt::read_lane(ns.active, 0);

    // R = FFT(ref), kept in registers for both dither pairs.  (Issuing a staging step's loads one step ahead --
    // the first pair's under the reference's transform, the second's under the first's -- was measured 6 %
    // SLOWER: the loaded chunks sit in registers across a transform and the allocator pays in accumulator-file
    // moves, profiles/r03/disp5_early_issue_ab.txt; each step issues and commits its own loads.)
    cf rh[8][8];
    {
        Stage5Loads<FOLD, 0, TIn> lref;
        issue5<FOLD, 0, TIn>(lref, ref, ref, ny, nx);
        commit5<FOLD, 0, TIn>(lds, lref, ns);
    }
    rt::block_sync_lds();
    forward5<FOLD, true>(lds, rh);

    float bv = -__builtin_inff();
    int bi = 0x7fffffff;
    const int NX = 2 * nx, NY = 2 * ny;
    const float oscale = 1.0f / (float)(L::P * L::P);
    for (int pr = 0; pr < 2; ++pr) {         // (00, 10) -> interlaced rows 2 qy, (01, 11) -> rows 2 qy + 1
        cf v[8][8];
        // the staging area is the exchange region: every wave must be past its last transposition (the
        // reference's forward transform, or the previous pair's) and past the interlace that read the planes
        rt::block_sync_lds();
        {
            Stage5Loads<FOLD, 1, TIn> lpair;
            issue5<FOLD, 1, TIn>(lpair, im4 + (2 * pr) * stride, im4 + (2 * pr + 1) * stride, ny, nx);
            commit5<FOLD, 1, TIn>(lds, lpair, ns);
        }
        rt::block_sync_lds();
        forward5<FOLD, false>(lds, v);
#pragma unroll
        for (int r = 0; r < 64; ++r) v[r >> 3][r & 7] = cmul(v[r >> 3][r & 7], rh[r >> 3][r & 7]);
        inverse5<FOLD>(lds, v);
        interlace_pair<FOLD>(lds, ny, nx, oscale, icc, pr, bv, bi);
    }
    rt::block_sync();        // icc (GLOBAL memory) written above is read below by other waves
    block_argmax(scr, bv, bi, 0);
    const bool nonfinite = !(bv < __builtin_inff());      // see disp5_body
    const int jmax = bi / NX, imax = bi % NX;
    PeakResult pk;
    if (nonfinite) {
        pk.x = (double)imax; pk.y = (double)jmax; pk.status = ST_NONFINITE;
    } else {
        pk = peak_from_argmax(scr, imax, jmax, NX, NY, [&](int x, int y) { return icc[y * NX + x]; });
    }
    if (fresh_tid() == 0) {
        out[0] = 0.5 * pk.x - (double)((NX - 1) / 4);     // cc.py:89-93
        out[1] = 0.5 * pk.y - (double)((NY - 1) / 4);
        if (status) status[0] = pk.status;
    }
}

template <bool FOLD, typename TIn = float>
SPX_TKERNEL1(256) void disp5p_kernel(const TIn* __restrict__ ref, const TIn* __restrict__ im4,
                                     int64_t nbatch, int ny, int nx, int cc_type,
                                     const cf* __restrict__ tw_g, float* __restrict__ icc,
                                     double* __restrict__ out, int* __restrict__ status, ItemTable items) {
    SPX_DYN_LDS(lds);
    load_twiddles<2>(lds, tw_g);
    for (int64_t p = first_item(rt::block_id(), rt::grid_size()); p < nbatch; p += rt::grid_size()) {
        const ItemView it = item_view(items, p, ny, nx);
        if (!it.ok) { if (!it.skip) item_refused(out, status, p, rt::thread_id() == 0); continue; }
        disp5p_body<FOLD, TIn>(ref + it.off, im4 + 4 * it.off, it.ny, it.nx, cc_type,
                               icc + 4 * it.off, out + 2 * p, status ? status + p : nullptr, lds);
        rt::block_sync_lds();
    }
}

}  // namespace p5
}  // namespace spx
