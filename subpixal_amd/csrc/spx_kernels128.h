// spx_kernels128.h -- cutouts of 86..128 pixels per side (BASELINE.json config 3 is 128x128): FFT period
// P = 192 = 3 x 64 (any P > 2n - 2 - (n-1)/2 keeps the reference's 'same' window free of aliasing;
// DESIGN.md section 1).  Same path as spx_kernels.h with C = P / 64 = 3 classes per axis (the templates
// also carry C = 4, period 256, which round 1 used above 96 px; only C = 3 is instantiated now).
//
// The padded spectrum splits into C*C classes  Z[C k'+c] = FFT64{ fold_c(z)[x'] w_P^(c x') },
// fold_c(z)[x'] = z[x'] + w_C^c z[x'+64], w_C = exp(-2 pi i / C)  (per axis; the cutout spans at
// most two 64-blocks), each again a 64x64 complex FFT done by one wave in registers with the
// machinery of spx_kernels.h.  A workgroup (4 waves) runs C rounds of C classes (for C = 3 the
// fourth wave helps staging and otherwise shadows class (cy, 0) without storing anything).  The
// per-class results  g_c[l'] = sum_{k in c} Z[k]^2 e^{2 pi i k l'/P}  (complex, 64x64) do not fit
// the LDS next to a second workgroup ("stresses LDS tile sizing"): the first two rounds' planes go
// to a per-workgroup workspace, the last round's stay in the exchange buffers; a radix-C pass
// combines them into the full PxP real convolution  conv[l'+64s] = Im( sum_c conj(w_C)^(c.s) g_c[l'] ) / 2P^2
// in the workspace, and the arg-max, the MFMA refine (period-P Dirichlet kernel, real, float64) and
// the fit read from it.
//
// Workspace per workgroup: C*C x 2 planes of 64x64 floats + P x (P+4) floats: 435 KiB (C = 3).
#pragma once

namespace spx {

template <int C> struct LdsBig {
    static constexpr int P = 64 * C;
    static constexpr int ZS = 72, XS = 68;
    static constexpr int TW_OFF = 0;                       // cf[P]
    static constexpr int SCR_OFF = TW_OFF + 256 * 8;       // 1 KiB scratch (table slot sized for P = 256)
    static constexpr int R_OFF = SCR_OFF + 1024;
    static constexpr int ZBUF_BYTES = 2 * 64 * ZS * 4;
    static constexpr int XCH_WAVE_BYTES = 64 * XS * 4;
    static constexpr int XCH_BYTES = 4 * XCH_WAVE_BYTES;
    // the 4 per-wave fine windows reuse the exchange region, which is idle during the
    // arg-max / refine / fit tail: 72.7 KiB in all -> two workgroups per CU
    static constexpr int FB_OFF = R_OFF;
    static constexpr int total(int W) {
        return R_OFF + (XCH_BYTES > 4 * W * W * 4 ? XCH_BYTES : 4 * W * W * 4);
    }
    // rows of the PxP convolution carry 4 extra columns that repeat columns 0..3, so that any
    // 4 consecutive (circular) columns are contiguous in memory (fine_window_big's 16-byte loads)
    static constexpr int CS = P + 4;
    static constexpr size_t kPlaneFloats = 64 * 64;
    static constexpr size_t kConvOffsetFloats = (size_t)C * C * 2 * kPlaneFloats;
    static constexpr size_t kWsBytes = (kConvOffsetFloats + (size_t)P * CS) * sizeof(float);
    // x mod P for x in [-P, 2P)
    static SPX_DEVICE int wrap(int x) {
        x += x < 0 ? P : 0;
        x -= x >= P ? P : 0;
        // callers stay inside [-P, 2P); anything else (a corrupted index) is folded into range
        // instead of becoming an out-of-bounds address
        return (unsigned)x < (unsigned)P ? x : 0;
    }
};
typedef LdsBig<4> Lds128;
constexpr size_t kWs128PlaneFloats = 64 * 64;
constexpr size_t kWs128Bytes = LdsBig<4>::kWsBytes;
constexpr size_t kWs96Bytes = LdsBig<3>::kWsBytes;
struct __attribute__((packed, aligned(4))) F32x4U { float v[4]; };   // 16-byte load, 4-byte aligned

// Staging of one round (all classes (cy, *)): the radix-C fold ALONG Y of z = ref + i*bal*flip(img)
//   u_cy[y'][x] = z[y'][x] + w_C^cy z[y'+64][x],   y' in [0,64), x in [0,128)
// (the cutout spans at most two 64-blocks per axis), normalised, into two LDS planes of 64 rows x
// US floats that together fill the exchange region exactly.  One pass over the pair per round:
// every thread issues its 16-byte loads back to back (one exposed memory latency per half),
// then folds and stores.  The waves then fold along x themselves: v = u[x'] + w_C^cx u[x'+64].
constexpr int kUS = 136;         // row stride (floats): 128 + 8, conflict-free 8x8 tile reads
template <int C, typename TIn, bool NARROW, bool NX4 = false>
SPX_DEVICE void stage_yfold_rows(unsigned char* lds, const TIn* __restrict__ ref,
                            const TIn* __restrict__ img, int ny, int nx,
                            const NormStatsT<TIn>& ns, float bal, int cy, float (&ssq)[2]) {
    typedef LdsBig<C> L;
    static_assert(2 * 64 * kUS * 4 <= L::XCH_BYTES, "the folded slab must fit the exchange region");
    const int tid = fresh_tid();
    float* ure = reinterpret_cast<float*>(lds + L::R_OFF);
    float* uim = ure + 64 * kUS;
    // w_C^cy = exp(-2 pi i cy / C)
    const float wr = cy == 0 ? 1.0f : (C == 4 ? (cy == 2 ? -1.0f : 0.0f) : -0.5f);
    const float wi = cy == 0 ? 0.0f : (C == 4 ? (cy == 1 ? -1.0f : (cy == 3 ? 1.0f : 0.0f))
                                              : (cy == 1 ? -0.86602540378443865f : 0.86602540378443865f));
    // 64 rows x 32 chunks = 2048 chunk pairs (top row y', bottom row y'+64), 8 per thread, in two
    // batches of 4: the 16 loads of a batch are in flight together (chunk_issue is branch-free)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        ChunkLoad<TIn> top[4], bot[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + (4 * half + i) * kThreads;
            const int yl = idx >> 5, x = (idx & 31) << 2;
            top[i] = chunk_issue<TIn, NARROW, NX4>(ref, img, ny, nx, yl, x);
            bot[i] = chunk_issue<TIn, NARROW, NX4>(ref, img, ny, nx, yl + 64, x);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + (4 * half + i) * kThreads;
            const int yl = idx >> 5, x = (idx & 31) << 2;
            float tre[4], tim[4], bre[4], bim[4];
            chunk_unpack(top[i], ns, tre, tim);
            chunk_unpack(bot[i], ns, bre, bim);
            if (cy == 0) {                        // sums of squares of the staged pixels (balance factor: round 0)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ssq[0] += tre[e] * tre[e] + bre[e] * bre[e];
                    ssq[1] += tim[e] * tim[e] + bim[e] * bim[e];
                }
            }
            f32x4 ore, oim;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // top + w * bottom (w = 1: exact sums); the image carries the balance factor
                const float ti = bal * tim[e], bi = bal * bim[e];
                ore[e] = __builtin_fmaf(-wi, bi, __builtin_fmaf(wr, bre[e], tre[e]));
                oim[e] = __builtin_fmaf(wi, bre[e], __builtin_fmaf(wr, bi, ti));
            }
            *reinterpret_cast<f32x4*>(ure + yl * kUS + x) = ore;
            *reinterpret_cast<f32x4*>(uim + yl * kUS + x) = oim;
        }
    }
}

// Full 128x128 float32 tiles (BASELINE config 3), 16-byte aligned: plain 16-byte loads, ALL 32 of a thread's
// loads of the round in flight together (the general routine above keeps 16: its chunk records need twice
// the registers) -- one exposed trip to memory per round instead of two.
template <int C>
SPX_DEVICE void stage_yfold_full128(unsigned char* lds, const float* __restrict__ ref,
                                    const float* __restrict__ img, const NormStatsT<float>& ns, float bal,
                                    int cy, float (&ssq)[2]) {
    typedef LdsBig<C> L;
    const int tid = fresh_tid();
    float* ure = reinterpret_cast<float*>(lds + L::R_OFF);
    float* uim = ure + 64 * kUS;
    const float wr = cy == 0 ? 1.0f : (C == 4 ? (cy == 2 ? -1.0f : 0.0f) : -0.5f);
    const float wi = cy == 0 ? 0.0f : (C == 4 ? (cy == 1 ? -1.0f : (cy == 3 ? 1.0f : 0.0f))
                                              : (cy == 1 ? -0.86602540378443865f : 0.86602540378443865f));
    const f32x4* r4 = reinterpret_cast<const f32x4*>(ref);
    const f32x4* m4 = reinterpret_cast<const f32x4*>(img);
    f32x4 rt[8], rb[8], mt[8], mb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + i * kThreads;                 // 64 rows x 32 chunks
        const int yl = idx >> 5, c = idx & 31;
        rt[i] = r4[yl * 32 + c];
        rb[i] = r4[(yl + 64) * 32 + c];
        mt[i] = m4[(127 - yl) * 32 + (31 - c)];             // flip(img): row 127 - y, columns back to front
        mb[i] = m4[(63 - yl) * 32 + (31 - c)];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int idx = tid + i * kThreads;
        const int yl = idx >> 5, x = (idx & 31) << 2;
        f32x4 ore, oim;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float tre = rt[i][e], bre = rb[i][e], tim = mt[i][3 - e], bim = mb[i][3 - e];
            if (ns.active) {
                tre = norm_ref(tre, ns); bre = norm_ref(bre, ns);
                tim = norm_im(tim, ns); bim = norm_im(bim, ns);
            }
            if (cy == 0) {
                ssq[0] += tre * tre + bre * bre;
                ssq[1] += tim * tim + bim * bim;
            }
            const float ti = bal * tim, bi = bal * bim;
            ore[e] = __builtin_fmaf(-wi, bi, __builtin_fmaf(wr, bre, tre));
            oim[e] = __builtin_fmaf(wi, bre, __builtin_fmaf(wr, bi, ti));
        }
        *reinterpret_cast<f32x4*>(ure + yl * kUS + x) = ore;
        *reinterpret_cast<f32x4*>(uim + yl * kUS + x) = oim;
    }
}

template <int C, typename TIn>
SPX_DEVICE void stage_yfold(unsigned char* lds, const TIn* __restrict__ ref,
                            const TIn* __restrict__ img, int ny, int nx,
                            const NormStatsT<TIn>& ns, float bal, int cy, float (&ssq)[2]) {
    if constexpr (sizeof(TIn) == 4) {
        if (ny == 128 && nx == 128 &&
            ((reinterpret_cast<uintptr_t>(ref) | reinterpret_cast<uintptr_t>(img)) & 15) == 0) {
            stage_yfold_full128<C>(lds, ref, img, ns, bal, cy, ssq);
            return;
        }
    }
    // cutouts narrower than a load chunk: element loads (chunk_issue), uniform per item
    if (nx < 4) stage_yfold_rows<C, TIn, true>(lds, ref, img, ny, nx, ns, bal, cy, ssq);
    else if ((nx & 3) == 0) stage_yfold_rows<C, TIn, false, true>(lds, ref, img, ny, nx, ns, bal, cy, ssq);
    else stage_yfold_rows<C, TIn, false>(lds, ref, img, ny, nx, ns, bal, cy, ssq);
}

// Round 0 (cy = 0) also finds the balance factor: its fold along y has w = 1, so the slab keeps
// ref and image apart (real plane = ref, imaginary plane = image) and the factor -- known only
// once the whole pair has been staged -- is applied when the waves read the slab; the later
// rounds stage with it.  No separate pass over the pair for the sums of squares.
template <int C, int DBG, typename TIn>
SPX_DEVICE void class_round128(unsigned char* lds, const TIn* __restrict__ ref,
                               const TIn* __restrict__ img, int ny, int nx, const NormStatsT<TIn>& ns,
                               float& bal, int cy, float* __restrict__ ws, PhaseClock<DBG>& clk) {
    typedef LdsBig<C> L;
    static_assert(C == 3 || C == 4, "");
    const int tid = fresh_tid();
    const int wave = rt::read_lane(tid >> 6, 0), lane = tid & 63;      // (scalar: the branches on it are uniform)
    // C = 3: the fourth wave has no class of its own.  It used to run class (cy, 0) along, storing nothing;
    // since round 3 it skips the transforms (SPX_IDLE_WAVE_SKIPS, a wave-uniform branch) and leaves its SIMD's
    // issue slots to the other workgroup's waves.
    const bool active = C == 4 || wave < C;
    const int cx = active ? wave : 0;
    const int l1 = lane >> 3, l0 = lane & 7;
    const cf* tw = reinterpret_cast<const cf*>(lds + L::TW_OFF);
    const float* zre = reinterpret_cast<const float*>(lds + L::R_OFF);
    float* xch = reinterpret_cast<float*>(lds + L::R_OFF + wave * L::XCH_WAVE_BYTES);

    rt::block_sync_lds();      // the previous round's exchange buffers alias the staging area
    float ssq[2] = {0.0f, 0.0f};
    stage_yfold<C, TIn>(lds, ref, img, ny, nx, ns, cy == 0 ? 1.0f : bal, cy, ssq);
    float ib = 1.0f;           // factor still to be applied to the imaginary plane
    if (cy == 0) {
        bal = balance_factor(lds + L::SCR_OFF, ssq);      // includes the barrier after staging
        ib = bal;
    } else {
        rt::block_sync_lds();
    }
    // fold along x: v = u[y][x'] + w_C^cx u[y][x'+64]
    cf v[8][8];
    {
        const float* ure = zre;
        const float* uim = zre + 64 * kUS;
        const cf w = cf{cx == 0 ? 1.0f : (C == 4 ? (cx == 2 ? -1.0f : 0.0f) : -0.5f),
                        cx == 0 ? 0.0f : (C == 4 ? (cx == 1 ? -1.0f : (cx == 3 ? 1.0f : 0.0f))
                                                 : (cx == 1 ? -0.86602540378443865f : 0.86602540378443865f))};
        const bool two = nx > 64;                 // the right half is all padding otherwise
#pragma unroll
        for (int y1 = 0; y1 < 8; ++y1)
#pragma unroll
            for (int x1 = 0; x1 < 8; ++x1) {
                const int a = (l1 + 8 * y1) * kUS + l0 + 8 * x1;
                cf d = cf{ure[a], ib * uim[a]};
                if (two) rt::cmac_ip(d, cf{ure[a + 64], ib * uim[a + 64]}, w);
                v[y1][x1] = d;
            }
    }
    rt::block_sync_lds();      // every wave has read the slab before any transposition overwrites it
    clk.tick(1);
    rt::set_prio<0>();         // transforms: throughput work (see spx_rt_hip.h set_prio)
#ifndef SPX_IDLE_WAVE_SHADOWS
    if (active) {
#endif
    // class pre-twiddle w_P^{c (8 y1)}
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
    for (int y0 = 1; y0 < 8; ++y0) {          // y0 = 0: w^0
        const cf wy = tw[y0 * (cy + C * l1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[y0][j] = cmulc(v[y0][j], wy);
    }
#pragma unroll
    for (int x0 = 1; x0 < 8; ++x0) {          // x0 = 0: w^0
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
#ifndef SPX_IDLE_WAVE_SHADOWS
    }
#endif
    clk.tick(2);
    rt::set_prio<1>();         // stores, staging, combine, refine: latency-bound
    // g_c -> workspace planes [class][re|im][64][64], through the wave's LDS buffer so the
    // global stores are 16-byte, row-contiguous.  Two economies (C = 3):
    //  * the real plane of class (0,0) is never written: its twiddle is 1 for every block, so
    //    only Im g_00 reaches conv = Im(sum_c conj(w)^(c.s) g_c);
    //  * the LAST round's planes stay on chip -- nothing overwrites the exchange region until the
    //    combine has read them: each wave leaves its real plane in its own buffer, wave 0 parks
    //    its imaginary plane in the idle fourth wave's buffer (4 of the 18 planes, 128 KiB of
    //    workspace traffic per pair less); the other two imaginary planes go to the workspace.
    const bool park = C == 3 && cy == C - 1;
    if (park) rt::block_sync_lds();            // every wave is past its last transposition
    float* g = ws + (size_t)((cy * C + cx) * 2) * kWs128PlaneFloats;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int part = park ? 1 - k : k;     // parking: imaginary first, the real plane stays
        const bool unused = C == 3 && cy == 0 && cx == 0 && part == 0;
        if (unused || (park && !active)) continue;
#pragma unroll
        for (int y1 = 0; y1 < 8; ++y1)
#pragma unroll
            for (int x1 = 0; x1 < 8; ++x1)
                xch[plane_elem(l1 + 8 * y1, l0 + 8 * x1)] = part ? v[y1][x1].y : v[y1][x1].x;
        rt::wave_sync();
        if (active && park && part == 1 && wave == 0) {      // LDS -> LDS: the idle wave's buffer
            f32x4* dst = reinterpret_cast<f32x4*>(lds + L::R_OFF + 3 * L::XCH_WAVE_BYTES);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                dst[i * 64 + lane] = reinterpret_cast<const f32x4*>(xch)[i * 64 + lane];
        } else if (active && !(park && part == 0)) {
            // (laundered: keeps the GLOBAL address space, or the two branches are merged into flat stores)
            f32x4* dst = rt::launder_lanes(reinterpret_cast<f32x4*>(g + part * kWs128PlaneFloats));
#pragma unroll
            for (int i = 0; i < 16; ++i)
                dst[i * 64 + lane] = reinterpret_cast<const f32x4*>(xch)[plane_slot(i * 64 + lane)];
        }
        rt::wave_sync();
    }
    clk.tick(3);
}

// DFT-C with the inverse transform's sign along one class axis:  X[s] = sum_c conj(w_C)^(c s) a[c]
template <int C> SPX_DEVICE void class_dft(const cf (&a)[C], cf (&x)[C]) {
    if constexpr (C == 4) {
        const cf e0 = a[0] + a[2], e1 = a[0] - a[2], o0 = a[1] + a[3], o1 = a[1] - a[3];
        x[0] = e0 + o0;
        x[2] = e0 - o0;
        x[1] = rt::add_pi(e1, o1);      // e1 + i o1
        x[3] = rt::add_mi(e1, o1);      // e1 - i o1
    } else {
        // conj(w_3) = -1/2 + i sqrt(3)/2:  X1,2 = a0 - (a1 + a2)/2 +- i (sqrt(3)/2)(a1 - a2)
        const cf t1 = a[1] + a[2], t2 = a[1] - a[2];
        const cf m = cf{a[0].x - 0.5f * t1.x, a[0].y - 0.5f * t1.y};
        const cf n = cf{-0.86602540378443865f * t2.y, 0.86602540378443865f * t2.x};
        x[0] = a[0] + t1;
        x[1] = m + n;
        x[2] = m - n;
    }
}

// radix-C combination of the C*C class planes into the full real convolution:
// conv[l'+64 s] = out_scale * Im( sum_c conj(w_C)^(cy sy + cx sx) g_c[l'] )
// fused with what the caller needs from the flipped 'same' window (rows [loy, loy+ny) x columns
// [lox, lox+nx) of the convolution; q = (n-1) + lo - l, conv_index inverted):
//   MODE 0 (pair mode):      the full convolution goes to `conv` (the refine stage reads all of it)
//                            and (bv, bi) take this thread's arg-max over the window, index qy*nx+qx;
//   MODE 1 (reference mode): the window goes straight to its interlaced positions
//                            icc[2 qy + oy][2 qx + ox] (cc.py:121-126), nothing else is stored, and
//                            (bv, bi) ACCUMULATE the arg-max over the interlaced image (NaN ranked
//                            as +inf, see nan_as_inf).
template <int C, int MODE>
SPX_DEVICE void combine128(const unsigned char* lds, const float* __restrict__ ws,
                           float* __restrict__ conv, float out_scale,
                           int ny, int nx, float* __restrict__ icc, int ox, int oy, float& bv, int& bi) {
    typedef LdsBig<C> L;
    const int tid = fresh_tid();
    const int loy = (ny - 1) / 2, lox = (nx - 1) / 2;
    const int NX2 = 2 * nx;
#pragma unroll 2
    for (int i4 = tid; i4 < 64 * 64 / 4; i4 += kThreads) {          // 4 consecutive l'x per step
        f32x4 gre[C * C], gim[C * C];
#pragma unroll
        for (int c = 0; c < C * C; ++c) {
            // C = 3: planes of the last round that class_round128 left in LDS (real planes in the
            // waves' own buffers, the imaginary plane of class (2,0) in the fourth buffer); the real
            // plane of class (0,0) does not exist (it cannot reach the imaginary part of the sum)
            // (separate statements per address space: a pointer chosen by ?: would be a generic one)
            if (C == 3 && c == 0)
                gre[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            else if (C == 3 && c >= C * (C - 1))
                gre[c] = reinterpret_cast<const f32x4*>(lds + L::R_OFF + (c - C * (C - 1)) * L::XCH_WAVE_BYTES)[plane_slot(i4)];
            else
                gre[c] = reinterpret_cast<const f32x4*>(ws + (size_t)(c * 2) * kWs128PlaneFloats)[i4];
            if (C == 3 && c == C * (C - 1))
                gim[c] = reinterpret_cast<const f32x4*>(lds + L::R_OFF + 3 * L::XCH_WAVE_BYTES)[plane_slot(i4)];
            else
                gim[c] = reinterpret_cast<const f32x4*>(ws + (size_t)(c * 2 + 1) * kWs128PlaneFloats)[i4];
        }
        const int ly = i4 >> 4, lx = (i4 & 15) << 2;
        f32x4 o[C][C];                                               // [sy][sx]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cf h[C][C];                                              // [cy][sx]
#pragma unroll
            for (int cy = 0; cy < C; ++cy) {
                cf a[C];
#pragma unroll
                for (int cx = 0; cx < C; ++cx) a[cx] = cf{gre[cy * C + cx][e], gim[cy * C + cx][e]};
                class_dft<C>(a, h[cy]);
            }
#pragma unroll
            for (int sx = 0; sx < C; ++sx) {
                cf a[C], x[C];
#pragma unroll
                for (int cy = 0; cy < C; ++cy) a[cy] = h[cy][sx];
                class_dft<C>(a, x);
#pragma unroll
                for (int sy = 0; sy < C; ++sy) o[sy][sx][e] = out_scale * x[sy].y;
            }
        }
#pragma unroll
        for (int sy = 0; sy < C; ++sy) {
            const int qy = (ny - 1) + loy - (ly + 64 * sy);
            const bool rowin = qy >= 0 && qy < ny;
#pragma unroll
            for (int sx = 0; sx < C; ++sx) {
                if constexpr (MODE == 0) {
                    float* row = conv + (size_t)(ly + 64 * sy) * L::CS;
                    *reinterpret_cast<f32x4*>(row + lx + 64 * sx) = o[sy][sx];
                    if (sx == 0 && lx == 0) *reinterpret_cast<f32x4*>(row + L::P) = o[sy][sx];   // wrap copy
                }
                if (rowin) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int qx = (nx - 1) + lox - (lx + e + 64 * sx);
                        if (qx >= 0 && qx < nx) {
                            if constexpr (MODE == 0) {
                                const int idx = qy * nx + qx;
                                if (better(o[sy][sx][e], idx, bv, bi)) { bv = o[sy][sx][e]; bi = idx; }
                            } else {
                                const int gi = (2 * qy + oy) * NX2 + 2 * qx + ox;
                                const float val = o[sy][sx][e];
                                icc[gi] = val;
                                if (better(nan_as_inf(val), gi, bv, bi)) { bv = nan_as_inf(val); bi = gi; }
                            }
                        }
                    }
                }
            }
        }
    }
}

// cutout pair -> class planes -> combine (ends with a full barrier).  MODE as in combine128.
template <int C, int DBG, typename TIn, int MODE>
SPX_DEVICE void conv_full128(unsigned char* lds, const TIn* __restrict__ ref,
                             const TIn* __restrict__ img, int ny, int nx, const NormStatsT<TIn>& ns,
                             float* __restrict__ ws, PhaseClock<DBG>& clk, float* __restrict__ icc,
                             int ox, int oy, float& bv, int& bi) {
    typedef LdsBig<C> L;
    float bal = 1.0f;                    // set by round 0
    clk.tick(0);
    for (int cy = 0; cy < C; ++cy) class_round128<C, DBG, TIn>(lds, ref, img, ny, nx, ns, bal, cy, ws, clk);
    rt::block_sync();                    // class planes (global) visible to every wave
    // conv = Im(IFFT(Z^2)) / 2, IFFT normalisation 1/P^2, balance undone
    const float out_scale = 0.5f / ((float)(L::P) * (float)(L::P) * bal);
    combine128<C, MODE>(lds, ws, ws + L::kConvOffsetFloats, out_scale, ny, nx, icc, ox, oy, bv, bi);
    rt::block_sync();
    clk.tick(4);
}

template <int C>
SPX_DEVICE float window_value128(const float* __restrict__ conv, int ny, int nx, int qy, int qx) {
    return conv[(size_t)conv_index(ny, qy) * LdsBig<C>::CS + conv_index(nx, qx)];
}

// Fine window around flipped coarse index (qyc, qxc) by MFMA, period-P real kernel
//   K(t) = 1/P [1 + 2 sum_{j=1..P/2-1} cos(2 pi j t / P) + cos(pi t)].
// Each of the four waves contracts all P rows for CW = P/4 window columns mx'' in
// [CW w - P/2, CW w - P/2 + CW) (TPW = CW/16 column tiles: 3 for P = 192, 4 for P = 256) and
// leaves its partial window in its own LDS buffer; the reader adds the four (fine_value128).
// Inside the wave, A-row lj of column tile t is column CW w + TPW lj + t, so that a lane's tiles
// are TPW consecutive columns = ONE 12- or 16-byte load per row; accumulator register r of tile t
// is then column CW w + TPW (lk + 4 r) + t.  Tables (spx_tables.h make_ktab_big), lane = 16 lk + lj:
//   [0][blk][lane][s]       = K(-(16 blk + lj - W/2)/U - (4 s + lk - P/2)),       s in [0, P/4)
//   [1][blk][lane][4 T + r] = K(-(16 blk + lj - W/2)/U - (CW w + TPW (lk + 4 r) + t - P/2)), T = TPW w + t
// (make_ktab_big_f64).
// The contractions accumulate in float64 (v_mfma_f64_16x16x4_f64, float64 tables): a float32
// chain over the P rows loses ~1e-7 of the peak value, which on a fine grid 20-60x flatter than
// the pixel grid is worth up to 2e-3 px at upsample >= 27; in float64 the float32 transforms
// themselves are the limit (~1e-5 px).  Its C/D map has row = lk + 4 r.
template <int TPW> struct __attribute__((packed, aligned(4))) RowFrag { float v[TPW]; };
struct __attribute__((aligned(16))) F64x2 { double v[2]; };
template <int C, int WB>
SPX_DEVICE void fine_window128(unsigned char* lds, const double* __restrict__ ktab,
                               const float* __restrict__ conv, int ny, int nx, int qyc, int qxc) {
    typedef LdsBig<C> L;
    typedef rt::f64x4 f64x4;
    constexpr int W = 16 * WB;
    constexpr int NQ = L::P / 16;            // groups of 4 table entries per (block, lane)
    constexpr int CW = L::P / 4;             // window columns per wave
    constexpr int TPW = CW / 16;             // column tiles per wave
    const int tid = fresh_tid();
    const int wave = tid >> 6, lane = tid & 63;
    const int lk = lane >> 4, lj = lane & 15;
    float* fbuf = reinterpret_cast<float*>(lds + L::FB_OFF) + wave * W * W;
    // (window centre clamped into the cutout: a corrupted index must not become an address)
    qyc = qyc < 0 ? 0 : (qyc > ny ? ny : qyc);
    qxc = qxc < 0 ? 0 : (qxc > nx ? nx : qxc);
    const int lyc = conv_index(ny, qyc), lxc = conv_index(nx, qxc);
    ktab = rt::launder(ktab);
    const F64x2* kty = reinterpret_cast<const F64x2*>(ktab) + (size_t)lane * NQ * 2;
    const F64x2* ktx = reinterpret_cast<const F64x2*>(ktab) + (size_t)(WB * 64 + lane) * NQ * 2;

    f64x4 acc[WB][TPW];
#pragma unroll
    for (int ab = 0; ab < WB; ++ab)
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[ab][t] = f64x4{0., 0., 0., 0.};
    const int col0 = L::wrap(lxc + CW * wave + TPW * lj - L::P / 2);     // + t (wrap copy in the row)
    // the convolution rows come from the workspace (Infinity Cache at best: ~600 cycles): the loads of step
    // s4 + 1 are issued before the matrix products of step s4
    RowFrag<TPW> nxt[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        nxt[e] = *reinterpret_cast<const RowFrag<TPW>*>(conv + (size_t)L::wrap(lyc + 4 * e + lk - L::P / 2) * L::CS + col0);
#pragma unroll 4
    for (int s4 = 0; s4 < NQ; ++s4) {
        double kb[WB][4];
#pragma unroll
        for (int ab = 0; ab < WB; ++ab) {
            const F64x2 lo = kty[((size_t)ab * 64 * NQ + s4) * 2], hi = kty[((size_t)ab * 64 * NQ + s4) * 2 + 1];
            kb[ab][0] = lo.v[0]; kb[ab][1] = lo.v[1]; kb[ab][2] = hi.v[0]; kb[ab][3] = hi.v[1];
        }
        RowFrag<TPW> a4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) a4[e] = nxt[e];
        if (s4 + 1 < NQ) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = L::wrap(lyc + 4 * (4 * (s4 + 1) + e) + lk - L::P / 2);
                nxt[e] = *reinterpret_cast<const RowFrag<TPW>*>(conv + (size_t)row * L::CS + col0);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < TPW; ++t)
#pragma unroll
                for (int ab = 0; ab < WB; ++ab)
                    acc[ab][t] = rt::mfma_f64_16x16x4((double)a4[e].v[t], kb[ab][e], acc[ab][t]);
    }
    // stage 2, one block of fine x offsets at a time (keeps the accumulators of large windows
    // within the register file)
#pragma unroll
    for (int bb = 0; bb < WB; ++bb) {
        f64x4 f[WB];
#pragma unroll
        for (int ab = 0; ab < WB; ++ab) f[ab] = f64x4{0., 0., 0., 0.};
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const size_t q = ((size_t)bb * 64 * NQ + TPW * wave + t) * 2;
            const F64x2 lo = ktx[q], hi = ktx[q + 1];
            const double ka[4] = {lo.v[0], lo.v[1], hi.v[0], hi.v[1]};
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ab = 0; ab < WB; ++ab)
                    f[ab] = rt::mfma_f64_16x16x4(ka[r], acc[ab][t][r], f[ab]);
        }
#pragma unroll
        for (int ab = 0; ab < WB; ++ab)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                fbuf[(bb * 16 + lk + 4 * r) * W + ab * 16 + lj] = (float)f[ab][r];
    }
    rt::block_sync_lds();
}

template <int W> SPX_DEVICE float fine_value128(const unsigned char* lds, int b, int a) {
    const float* fbuf = reinterpret_cast<const float*>(lds + LdsBig<4>::FB_OFF);   // same offset for every C
    float acc = fbuf[b * W + a];
#pragma unroll
    for (int w = 1; w < 4; ++w) acc += fbuf[w * W * W + b * W + a];
    return acc;
}

// ---------------------------------------------------------------------------
// pair mode, 128 tile
// ---------------------------------------------------------------------------
template <int C, int WB, int DBG, typename TIn>
SPX_DEVICE void pair128_body(const TIn* __restrict__ ref, const TIn* __restrict__ img, int ny,
                             int nx, int U, int cc_type, const double* __restrict__ ktab,
                             double* __restrict__ out, int* __restrict__ status,
                             unsigned char* lds, float* __restrict__ ws, PhaseClock<DBG>& clk) {
    typedef LdsBig<C> L;
    ny = rt::launder_uniform(ny);
    nx = rt::launder_uniform(nx);
    U = rt::launder_uniform(U);
    const int tid = fresh_tid();
    unsigned char* scr = lds + L::SCR_OFF;
    const NormStatsT<TIn> ns = norm_stats(scr, ref, img, 1, 0, ny, nx, cc_type);
    // coarse arg-max over the flipped 'same' window: found by the combine pass itself
    float bv = -__builtin_inff();
    int bi = kNoIndex;
    conv_full128<C, DBG, TIn, 0>(lds, ref, img, ny, nx, ns, ws, clk, nullptr, 0, 0, bv, bi);
    const float* conv = ws + L::kConvOffsetFloats;
    block_argmax(scr, bv, bi, 0);
    const bool nonfinite = bi == kNoIndex;       // NaN everywhere (see pair_body in spx_kernels.h)
    if (nonfinite) bi = 0;
    int qyc = bi / nx, qxc = bi - (bi / nx) * nx;
    clk.tick(5);
    PeakResult pk;
    if (nonfinite) {
        pk.x = 0.0; pk.y = 0.0; pk.status = ST_NONFINITE;
    } else if constexpr (WB == 0) {
        pk = peak_fit_wave0(scr, qxc, qyc, nx, ny, [&](int x, int y) {
            return window_value128<C>(conv, ny, nx, y, x);
        });
    } else {
        constexpr int W = 16 * (WB > 0 ? WB : 1);
        const int NX = U * nx, NY = U * ny;
        int imax = 0, jmax = 0;
        bool inside = false;
        for (int iter = 0; iter < 4; ++iter) {
            fine_window128<C, (WB > 0 ? WB : 1)>(lds, ktab, conv, ny, nx, qyc, qxc);
            clk.tick(6);
            const int fx0 = U * qxc - W / 2, fy0 = U * qyc - W / 2;
            float fv = -__builtin_inff();
            int fi = 0x7fffffff;
            for (int idx = tid; idx < W * W; idx += kThreads) {
                const int a = idx / W, b = idx % W;
                const int gy = fy0 + a, gx = fx0 + b;
                if (gy >= 0 && gy < NY && gx >= 0 && gx < NX) {
                    const float val = fine_value128<W>(lds, b, a);
                    if (better(val, idx, fv, fi)) { fv = val; fi = idx; }
                }
            }
            block_argmax(scr, fv, fi, 1);
            clk.tick(7);
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
            rt::block_sync_lds();       // everyone has read the window before it is rebuilt
        }
        if (inside) {
            const int fx0 = U * qxc - W / 2, fy0 = U * qyc - W / 2;
            pk = peak_fit_wave0(scr, imax, jmax, NX, NY, [&](int x, int y) {
                return fine_value128<W>(lds, x - fx0, y - fy0);
            });
        } else if (imax < 0) {
            pk.x = 0.0; pk.y = 0.0; pk.status = ST_NONFINITE;
        } else {
            pk.x = (double)imax; pk.y = (double)jmax; pk.status = ST_WINDOW;
        }
    }
    if (tid == 0) {
        out[0] = pk.x / (double)U - (double)((nx - 1) / 2);
        out[1] = pk.y / (double)U - (double)((ny - 1) / 2);
        if (status) status[0] = pk.status;
    }
    clk.tick(8);
}

template <int C> SPX_DEVICE void load_twiddles128(unsigned char* lds, const cf* __restrict__ tw_g) {
    cf* tw = reinterpret_cast<cf*>(lds + LdsBig<C>::TW_OFF);
    for (int i = rt::thread_id(); i < LdsBig<C>::P; i += kThreads) tw[i] = tw_g[i];
    rt::block_sync_lds();
}

template <int C, int WB, int DBG = 0, typename TIn = float>
SPX_TKERNEL(256) void pair128_kernel(const TIn* __restrict__ ref, const TIn* __restrict__ img,
                                     int64_t nbatch, int ny, int nx, int U, int cc_type,
                                     const cf* __restrict__ tw_g, const double* __restrict__ ktab,
                                     double* __restrict__ out, int* __restrict__ status,
                                     float* __restrict__ workspace) {
    SPX_DYN_LDS(lds);
    load_twiddles128<C>(lds, tw_g);
    float* ws = workspace + (size_t)rt::block_id() * (LdsBig<C>::kWsBytes / sizeof(float));
    const int64_t stride = (int64_t)ny * nx;
    PhaseClock<DBG> clk;
    clk.start();
    for (int64_t p = first_item(rt::block_id(), rt::grid_size()); p < nbatch; p += rt::grid_size()) {
        pair128_body<C, WB, DBG, TIn>(ref + p * stride, img + p * stride, ny, nx, U, cc_type, ktab, out + 2 * p,
                              status ? status + p : nullptr, lds, ws, clk);
        rt::block_sync();
        clk.tick(9);
    }
    // diagnostic build: the per-phase cycle totals go to the tail of the status array
    if constexpr (DBG == 100)
        clk.flush(reinterpret_cast<unsigned long long*>(status + nbatch));
}

// ---------------------------------------------------------------------------
// reference (5-image) mode, 96 / 128 tile
// ---------------------------------------------------------------------------
template <int C, typename TIn = float>
SPX_TKERNEL(256) void disp5_128_kernel(const TIn* __restrict__ ref, const TIn* __restrict__ im4,
                                       int64_t nbatch, int ny, int nx, int cc_type,
                                       const cf* __restrict__ tw_g, float* __restrict__ icc_all,
                                       double* __restrict__ out_all, int* __restrict__ status,
                                       float* __restrict__ workspace, ItemTable items) {
    typedef LdsBig<C> L;
    SPX_DYN_LDS(lds);
    load_twiddles128<C>(lds, tw_g);
    float* ws = workspace + (size_t)rt::block_id() * (L::kWsBytes / sizeof(float));
    unsigned char* scr = lds + L::SCR_OFF;
    const int ny_u = ny, nx_u = nx;
    for (int64_t p = first_item(rt::block_id(), rt::grid_size()); p < nbatch; p += rt::grid_size()) {
        const ItemView it = item_view(items, p, ny_u, nx_u);       // per-item shape (see spx_kernels.h)
        if (!it.ok) { if (!it.skip) item_refused(out_all, status, p, rt::thread_id() == 0); continue; }
        ny = it.ny;
        nx = it.nx;
        const int64_t stride = (int64_t)ny * nx;
        const int NX = 2 * nx, NY = 2 * ny;
        const TIn* r = ref + it.off;
        const TIn* m4 = im4 + 4 * it.off;
        float* icc = icc_all + 4 * it.off;
        const int tid = fresh_tid();
        const NormStatsT<TIn> ns = norm_stats(scr, r, m4, 4, stride, ny, nx, cc_type);
        float bv = -__builtin_inff();
        int bi = 0x7fffffff;
        PhaseClock<0> clk;
        for (int q = 0; q < 4; ++q) {
            const int ox = q & 1, oy = q >> 1;
            // the combine pass writes this dither's window straight into the interlaced image
            conv_full128<C, 0, TIn, 1>(lds, r, m4 + q * stride, ny, nx, ns, ws, clk, icc, ox, oy, bv, bi);
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
