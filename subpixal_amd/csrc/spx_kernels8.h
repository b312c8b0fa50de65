// spx_kernels8.h -- the 64 tile on EIGHT waves per pair (round 3): the same path as pair_kernel of
// spx_kernels.h (cc.py:114 fftconvolve -> cc.py:121-126 window -> centroid.py:114 arg-max ->
// upsampled refine -> centroid.py:158-236 fit), with the transform core rebuilt for occupancy.
//
// Why.  pair_kernel keeps one 64x64 complex class tile per wave: 128 VGPRs of tile, 253 in all, so two
// waves per SIMD -- and two waves cannot cover each other's LDS / barrier latency (69 % VALU-busy,
// profiles/r02).  Here a wave holds 32 complex registers (64 VGPRs of tile, <= 128 in all): FOUR waves
// per SIMD at the same LDS footprint per pair (two 512-thread workgroups per CU).
//
// How.  The zero padded 128-point transform is decimated by FOUR per axis instead of two:
//     Z[4k'+c] = FFT32{ w_P^(c x') sum_j (-i)^(c j) z[x' + 32 j] }[k'],   c in [0,4), x' in [0,32), j in {0,1}
// -- 16 classes (cy, cx) of 32x32 complex points.  1024 points = 32 lanes x 32 registers, so ONE class
// lives in HALF a wave and a 2-D class transform is, exactly as in pair_kernel, two register rounds with a
// single lane<->register transposition in between (5 bits <-> 5 bits, inside each half-wave):
//     round A   lane (ya, xa) in 4x8, registers (yb, xb) in 8x4; y' = ya + 4 yb, x' = xa + 8 xb
//               radix-8 over yb -> kyA, radix-4 over xb -> kxA; twiddle w_P^(ya (cy + 4 kyA) + xa (cx + 4 kxA))
//     transposition  (lane l5, register r5) -> (lane r5, register l5), through the wave's 8 KiB buffer
//     round B   registers (ya, xa): radix-4 over ya -> kyB, radix-8 over xa -> kxB
//               Z[cy + 4 (kyA + 8 kyB)][cx + 4 (kxA + 4 kxB)] on lane (kyA, kxA), register (kyB, kxB)
// W = Z^2 and the inverse runs backwards, leaving E_c[l'] = sum_{k = c mod 4} W[k] w_P^(-k l') on lane
// (ya, xa), register (yb, xb).  Wave w = 4 py + 2 cxl + e holds classes cy = py + 2 e, cx = cxl + 2 h
// (h = half-wave).  The four mod-4 classes behind one parity plane (py, cxl) recombine by a 2x2 butterfly:
//     d_p[l' + 32 j] = Im sum_{a,b} i^((py + 2a) jy + (cxl + 2b) jx) E_(py+2a, cxl+2b)[l'],   j in {0,1}^2
// over b between the half-waves (v_permlane32_swap, no LDS) and over a between the two waves of a plane,
// which trade ONE real value per register through their exchange buffers and then write the plane rows
// into each other's (now free) buffer: the four real parity planes d_p of pair_kernel, in its layout, are
// what the tail (arg-max, MFMA refine, fit) reads -- that part is pair_kernel's, on 512 threads.
//
// Needs spx_kernels.h first.  Compiled by hipcc for gfx950 and by the CPU logic-check harness.
#pragma once

namespace spx {
namespace w8 {

constexpr int kT8 = 512;           // threads per workgroup
constexpr int kW8 = 8;             // waves

struct L8 {
    static constexpr int P = 128;
    static constexpr int TW_OFF = 0;                      // cf[P]
    static constexpr int SCR_OFF = 1024;                  // 1 KiB scratch (S8_*)
    static constexpr int R_OFF = 2048;                    // staged input | exchange buffers | planes
    // staged z = ref + i bal flip(img): 64 rows x ZRS floats; within a row the 8 complex samples
    // x = xa + 8 m (m = 0..7) of lane-column xa are contiguous (ZXS floats per xa: 16 + 4 pad), so a lane's
    // 8 fold/radix inputs of a row are four 16-byte reads.  (ZXS, ZRS) = (20, 160): those reads and the
    // 8-byte staging writes are bank-conflict free (brute-forced).
    static constexpr int ZXS = 20, ZRS = 160;
    static constexpr int ZBUF_BYTES = 64 * ZRS * 4;
    static constexpr int XCH_WAVE_BYTES = 8192;           // 64 rows (h, r) x 32 floats, XOR-swizzled
    static constexpr int XCH_BYTES = kW8 * XCH_WAVE_BYTES;
    static constexpr int PLANE_BYTES = 2 * XCH_WAVE_BYTES;   // plane p = the buffers of waves 2p, 2p+1
    static constexpr int PS = 64;
    static constexpr int FB_OFF = R_OFF + XCH_BYTES;
    // fine window(s): one per wave for W = 16 (the reader adds the eight in fixed order), a single one
    // accumulated in wave order above that
    static constexpr int fb_count(int W) { return W <= 16 ? kW8 : 1; }
    static constexpr int total(int W) { return FB_OFF + fb_count(W) * W * W * 4; }
};
static_assert(L8::ZBUF_BYTES <= L8::XCH_BYTES, "staged input must fit the exchange region it shares");

// scratch sub-offsets
constexpr int S8_RED_F = 0;       // 2 slots x (float[8] + int[8])
constexpr int S8_STAT = 128;      // float[16]
constexpr int S8_RED_D6 = 256;    // double[8 * 6]

// sgn of a half-wave: +1 for lanes 0..31, -1 for lanes 32..63
SPX_DEVICE float half_sign(int lane) { return (lane & 32) ? -1.0f : 1.0f; }

// ---------------------------------------------------------------------------
// workgroup reductions over 8 waves
// ---------------------------------------------------------------------------
SPX_DEVICE void block_sum2f8(unsigned char* scr, float& a, float& b) {
    const int tid = rt::thread_id();
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        a += rt::shfl_xor(a, m);
        b += rt::shfl_xor(b, m);
    }
    float* part = reinterpret_cast<float*>(scr + S8_STAT);
    if ((tid & 63) == 0) { part[2 * (tid >> 6)] = a; part[2 * (tid >> 6) + 1] = b; }
    rt::block_sync_lds();
    a = ((part[0] + part[2]) + (part[4] + part[6])) + ((part[8] + part[10]) + (part[12] + part[14]));
    b = ((part[1] + part[3]) + (part[5] + part[7])) + ((part[9] + part[11]) + (part[13] + part[15]));
}

SPX_DEVICE void block_sum6d8(unsigned char* scr, double (&v)[6]) {
    const int tid = rt::thread_id();
    double* part = reinterpret_cast<double*>(scr + S8_RED_D6);
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = wave_sum(v[i]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) part[(tid >> 6) * 6 + i] = v[i];
    }
    rt::block_sync_lds();
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = 0.0;
        for (int w = 0; w < kW8; ++w) s += part[w * 6 + i];
        v[i] = s;
    }
    rt::block_sync_lds();
}

SPX_DEVICE void block_argmax8(unsigned char* scr, float& v, int& idx, int slot) {
    const int tid = rt::thread_id();
    wave_argmax(v, idx);
    float* rf = reinterpret_cast<float*>(scr + S8_RED_F) + 16 * slot;
    int* ri = reinterpret_cast<int*>(scr + S8_RED_F) + 16 * slot + 8;
    if ((tid & 63) == 0) { rf[tid >> 6] = v; ri[tid >> 6] = idx; }
    rt::block_sync_lds();
    v = rf[0];
    idx = ri[0];
    for (int w = 1; w < kW8; ++w)
        if (better(rf[w], ri[w], v, idx)) { v = rf[w]; idx = ri[w]; }
}

// cc.py:131-156 statistics (norm_stats of spx_kernels.h on 512 threads)
template <typename TIn>
SPX_DEVICE NormStatsT<TIn> norm_stats8(unsigned char* scr, const TIn* __restrict__ ref,
                                       const TIn* __restrict__ ims, int npool, int64_t im_stride,
                                       int ny, int nx, int cc_type) {
    NormStatsT<TIn> ns;
    ns.active = 0;
    ns.im_mean = 0; ns.im_rstd = 1; ns.ref_mean = 0; ns.ref_rstd = 1;
    if (cc_type == CC_PLAIN) return ns;
    const int tid = fresh_tid();
    const int npx = ny * nx;
    bool vec = (npx & 3) == 0 && (im_stride & 3) == 0 &&
               ((reinterpret_cast<uintptr_t>(ref) | reinterpret_cast<uintptr_t>(ims)) & 15) == 0;
    const int nchunk = vec ? npx >> 2 : npx;
    double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // n_im, sum_im, sum_im^2, n_union, sum_ref, sum_ref^2
#pragma unroll 2
    for (int i = tid; i < nchunk; i += kT8) {
        Quad<TIn> r4 = Quad<TIn>{{0, 0, 0, 0}};
        unsigned anym = 0;
        if (vec) r4 = load_quad(ref + 4 * (int64_t)i); else r4.v[0] = ref[i];
        for (int q = 0; q < npool; ++q) {
            Quad<TIn> m4 = Quad<TIn>{{0, 0, 0, 0}};
            if (vec) m4 = load_quad(ims + q * im_stride + 4 * (int64_t)i);
            else m4.v[0] = ims[q * im_stride + i];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (m4.v[e] != (TIn)0) {
                    const double x = (double)m4.v[e];
                    a[0] += 1.0; a[1] += x; a[2] += x * x; anym |= 1u << e;
                }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (anym & (1u << e)) {
                const double x = (double)r4.v[e];
                a[3] += 1.0; a[4] += x; a[5] += x * x;
            }
    }
    block_sum6d8(scr, a);
    const double n_im = a[0], n_un = a[3];
    const double im_mean = a[1] / n_im, ref_mean = a[4] / n_un;
    double b0 = a[2] - a[1] * im_mean, b1 = a[5] - a[4] * ref_mean;
    if (b0 < 0.0) b0 = 0.0;
    if (b1 < 0.0) b1 = 0.0;
    ns.active = 1;
    const bool zero = (cc_type == CC_ZNCC);
    ns.im_mean = zero ? (TIn)im_mean : (TIn)0;
    ns.im_rstd = (TIn)(1.0 / sqrt(b0 / n_im));
    ns.ref_mean = zero ? (TIn)ref_mean : (TIn)0;
    ns.ref_rstd = (TIn)(1.0 / sqrt(b1 / n_un));
    return ns;
}

// ---------------------------------------------------------------------------
// Staging: each thread fetches two 4-pixel chunks of ref and of the flipped image (registers), the
// workgroup agrees on the balance factor (one barrier), then z = ref + i bal flip(img) goes to LDS in the
// permuted complex layout of L8.  `bal` is applied here, not at the tile load, because the fold
// of the class inputs mixes real and imaginary parts.
// ---------------------------------------------------------------------------
template <typename TIn, bool NARROW, bool NX4>
SPX_DEVICE void fetch_pair8(const TIn* __restrict__ ref, const TIn* __restrict__ img, int ny, int nx,
                            const NormStatsT<TIn>& ns, float (&re)[2][4], float (&im)[2][4]) {
    const int tid = fresh_tid();
    const bool aligned = ((reinterpret_cast<uintptr_t>(ref) | reinterpret_cast<uintptr_t>(img)) & 15) == 0;
    if constexpr (sizeof(TIn) == 4 && !NARROW) if (ny == 64 && nx == 64 && aligned) {
        const f32x4* r4 = reinterpret_cast<const f32x4*>(ref);
        const f32x4* m4 = reinterpret_cast<const f32x4*>(img);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * kT8;                 // 1024 float4 per image
            const int y = idx >> 4;
            f32x4 r = r4[idx];
            const f32x4 t = m4[(63 - y) * 16 + (15 - (idx & 15))];
            f32x4 m = f32x4{t[3], t[2], t[1], t[0]};
            if (ns.active) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    m[e] = norm_im(m[e], ns);
                    r[e] = norm_ref(r[e], ns);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { re[i][e] = r[e]; im[i][e] = m[e]; }
        }
        return;
    }
    ChunkLoad<TIn> ld[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * kT8;
        ld[i] = chunk_issue<TIn, NARROW, NX4>(ref, img, ny, nx, idx >> 4, (idx & 15) << 2);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) chunk_unpack(ld[i], ns, re[i], im[i]);
}

template <typename TIn>
SPX_DEVICE float stage_pair8(unsigned char* lds, const TIn* __restrict__ ref, const TIn* __restrict__ img,
                             int ny, int nx, const NormStatsT<TIn>& ns) {
    float re[2][4], im[2][4];
    if (nx < 4) fetch_pair8<TIn, true, false>(ref, img, ny, nx, ns, re, im);          // (uniform per item)
    else if ((nx & 3) == 0) fetch_pair8<TIn, false, true>(ref, img, ny, nx, ns, re, im);
    else fetch_pair8<TIn, false, false>(ref, img, ny, nx, ns, re, im);
    float sr = 0.0f, sm = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { sr += re[i][e] * re[i][e]; sm += im[i][e] * im[i][e]; }
    block_sum2f8(lds + L8::SCR_OFF, sr, sm);        // (the previous pair's readers of this region are past their
                                                    //  last workgroup barrier: see pair8_kernel)
    const float bal = balance_from_ssq(sr, sm);
    const int tid = fresh_tid();
    float* zb = reinterpret_cast<float*>(lds + L8::R_OFF);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * kT8;
        const int y = idx >> 4, q = idx & 15;
        float* row = zb + y * L8::ZRS + (4 * (q & 1)) * L8::ZXS + (q >> 1) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            *reinterpret_cast<f32x2*>(row + e * L8::ZXS) = f32x2{re[i][e], bal * im[i][e]};
    }
    return bal;
}

// ---------------------------------------------------------------------------
// 32-register tile helpers
// ---------------------------------------------------------------------------
// radix-8 along the first digit of an [8][4] tile (index 4 a + b), radix-4 along the second
template <int DIR> SPX_DEVICE void fft_8x4(cf (&v)[32]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        cf t[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) t[a] = v[4 * a + b];
        fft8<DIR>(t);
#pragma unroll
        for (int a = 0; a < 8; ++a) v[4 * a + b] = t[a];
    }
#pragma unroll
    for (int a = 0; a < 8; ++a)
        fft4<DIR, false>(v[4 * a], v[4 * a + 1], v[4 * a + 2], v[4 * a + 3],
                         v[4 * a], v[4 * a + 1], v[4 * a + 2], v[4 * a + 3]);
}
// radix-4 along the first digit of a [4][8] tile (index 8 a + b), radix-8 along the second
template <int DIR> SPX_DEVICE void fft_4x8(cf (&v)[32]) {
#pragma unroll
    for (int b = 0; b < 8; ++b)
        fft4<DIR, false>(v[b], v[8 + b], v[16 + b], v[24 + b], v[b], v[8 + b], v[16 + b], v[24 + b]);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        cf t[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) t[b] = v[8 * a + b];
        fft8<DIR>(t);
#pragma unroll
        for (int b = 0; b < 8; ++b) v[8 * a + b] = t[b];
    }
}

// Lane <-> register transposition inside each half-wave: (lane l5, register r) -> (lane r, register l5),
// real and imaginary parts in two passes through the wave's 8 KiB buffer: 64 rows (h, r) of 32 floats.
// Logical 16-byte chunk j4 of row (h, r) is stored at chunk j4 ^ ((r >> 1) & 7): the writes (a whole row
// per half-wave and instruction) and the 16-byte reads (16 lanes = 16 rows per LDS cycle) are then both
// bank-conflict free without padding -- the buffers have to be exactly 8 KiB for the planes that follow.
SPX_DEVICE void transpose32(cf (&v)[32], float* xch, int lane) {
    const int h = lane >> 5, l5 = lane & 31;
    float* wr = xch + h * 1024;
    const float* rd = xch + h * 1024 + l5 * 32;
    const int key = (l5 >> 1) & 7;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int r = 0; r < 32; ++r)
            wr[r * 32 + (l5 ^ (((r >> 1) & 7) << 2))] = part ? v[r].y : v[r].x;
        rt::wave_sync();
#pragma unroll
        for (int j4 = 0; j4 < 8; ++j4) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(rd + ((j4 ^ key) << 2));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (part) v[4 * j4 + e].y = q[e]; else v[4 * j4 + e].x = q[e];
            }
        }
        rt::wave_sync();
    }
}

// [a_lo + a_hi | b_lo - b_hi] (values of lanes l and l+32 combined; the left form lands in lanes 0..31,
// the right one in lanes 32..63).  NEG: the right form negated.  One v_permlane32_swap + one FMA.
template <bool NEG> SPX_DEVICE float half_mix(float a, float b, float sgn) {
    rt::swap_halves(a, b);          // a = [a_lo, b_lo], b = [a_hi, b_hi]
    return NEG ? __builtin_fmaf(sgn, a, b) : __builtin_fmaf(sgn, b, a);
}

// ---------------------------------------------------------------------------
// Class inputs: u[yb][xb] = sum_{jy,jx} (-i)^(cy jy + cx jx) z[ya + 4 yb + 32 jy][xa + 8 xb + 32 jx]
// (cx = CXL + 2 h: the two half-waves differ by the sign `sgn` of the jx = 1 term).
// ---------------------------------------------------------------------------
template <int CY, int CXL>
SPX_DEVICE void fold_tile(const unsigned char* lds, int lane, cf (&v)[32]) {
    const int l5 = lane & 31, ya = l5 >> 3, xa = l5 & 7;
    const float s = half_sign(lane);
    const f32x2 sx = CXL ? f32x2{s, -s} : f32x2{s, s};
    const float* zb = reinterpret_cast<const float*>(lds + L8::R_OFF) + ya * L8::ZRS + xa * L8::ZXS;
#pragma unroll
    for (int yb = 0; yb < 8; ++yb) {
        cf t[2][4];
#pragma unroll
        for (int jy = 0; jy < 2; ++jy) {
            const f32x4* p = reinterpret_cast<const f32x4*>(zb + (4 * yb + 32 * jy) * L8::ZRS);
            const f32x4 c0 = p[0], c1 = p[1], c2 = p[2], c3 = p[3];       // samples m = 0..7 (x = xa + 8 m)
            const cf z0[4] = {cf{c0[0], c0[1]}, cf{c0[2], c0[3]}, cf{c1[0], c1[1]}, cf{c1[2], c1[3]}};
            const cf z1[4] = {cf{c2[0], c2[1]}, cf{c2[2], c2[3]}, cf{c3[0], c3[1]}, cf{c3[2], c3[3]}};
#pragma unroll
            for (int xb = 0; xb < 4; ++xb)      // z0 + (-i)^cx z1:  cx even: +-z1;  cx odd: +-(z1.y, -z1.x)
                t[jy][xb] = CXL ? rt::fma_swap(z1[xb], sx, z0[xb]) : rt::fma_pk(z1[xb], sx, z0[xb]);
        }
#pragma unroll
        for (int xb = 0; xb < 4; ++xb) {
            const cf a = t[0][xb], b = t[1][xb];
            v[4 * yb + xb] = CY == 0 ? a + b : CY == 1 ? rt::add_mi(a, b) : CY == 2 ? a - b : rt::add_pi(a, b);
        }
    }
}

// register part of the class twiddle: v[yb][xb] *= w_P^(4 cy yb + 8 cx xb)  (CONJ: the inverse's)
template <bool CONJ>
SPX_DEVICE void class_twiddle(const cf* tw, int cy, int cx, cf (&v)[32]) {
    if (cy) {                                   // (wave-uniform)
#pragma unroll
        for (int yb = 1; yb < 8; ++yb) {
            const cf w = tw[(4 * cy * yb) & 127];
#pragma unroll
            for (int xb = 0; xb < 4; ++xb) { if (CONJ) rt::cmulc_ip(v[4 * yb + xb], w); else rt::cmul_ip(v[4 * yb + xb], w); }
        }
    }
#pragma unroll
    for (int xb = 1; xb < 4; ++xb) {
        const cf w = tw[(8 * cx * xb) & 127];    // (cx differs between the half-waves)
#pragma unroll
        for (int yb = 0; yb < 8; ++yb) v[4 * yb + xb] = CONJ ? cmulc(v[4 * yb + xb], w) : cmul(v[4 * yb + xb], w);
    }
}

// Recombination over the half-waves.  T(jx) = sum_b i^(cx jx) E: lanes 0..31 take jx = 0 (D0 = E_lo + E_hi),
// lanes 32..63 jx = 1 (i^cxl D1, D1 = E_lo - E_hi).  The wave then owns Im T or Re T (`own`) and sends the
// other wave of its plane the part that one needs (`snd`):
//   E = 0 (cy = py):     rows 32.. :  PY = 0: Im Ta - Im Tb     PY = 1: Re Ta - Re Tb;   sends Im Ta
//   E = 1 (cy = py + 2): rows  0.. :  Im Ta + Im Tb;             sends Im Tb (PY = 0) / Re Tb (PY = 1)
template <int PY, int CXL, int E>
SPX_DEVICE void recombine8(const cf (&v)[32], int lane, float (&own)[32], float (&snd)[32]) {
    const float sgn = half_sign(lane);
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        // CXL = 0: [Im D0 | Im D1], [Re D0 | Re D1];  CXL = 1: [Im D0 | Im(i D1) = Re D1], [Re D0 | Re(i D1) = -Im D1]
        const float imT = CXL ? half_mix<false>(v[r].y, v[r].x, sgn) : half_mix<false>(v[r].y, v[r].y, sgn);
        float reT = 0.0f;
        if (PY) reT = CXL ? half_mix<true>(v[r].x, v[r].y, sgn) : half_mix<false>(v[r].x, v[r].x, sgn);
        if (E == 0) { snd[r] = imT; own[r] = PY ? reT : imT; }
        else { snd[r] = PY ? reT : imT; own[r] = imT; }
    }
}

// ---------------------------------------------------------------------------
// planes8: staged z -> the four real parity planes d_p in LDS (pair_kernel's cc_planes on 8 waves).
// The caller has issued a barrier after staging; ends with a barrier.  `role` = wave index + rotation.
// ---------------------------------------------------------------------------
template <int DBG>
SPX_DEVICE void planes8(unsigned char* lds, PhaseClock<DBG>& clk, int rot) {
    const int tid = fresh_tid();
    const int lane = tid & 63;
    const int role = rt::read_lane(((tid >> 6) + rot) & 7, 0);         // scalar
    const int py = role >> 2, cxl = (role >> 1) & 1, e = role & 1;
    const int cy = py + 2 * e;
    const int h = lane >> 5, l5 = lane & 31;
    const int cx = cxl + 2 * h;
    const cf* tw = reinterpret_cast<const cf*>(lds + L8::TW_OFF);
    float* xch = reinterpret_cast<float*>(lds + L8::R_OFF + role * L8::XCH_WAVE_BYTES);

    cf v[32];
    switch (2 * cy + cxl) {                    // (wave-uniform: one of eight instantiations per wave)
    case 0: fold_tile<0, 0>(lds, lane, v); break;
    case 1: fold_tile<0, 1>(lds, lane, v); break;
    case 2: fold_tile<1, 0>(lds, lane, v); break;
    case 3: fold_tile<1, 1>(lds, lane, v); break;
    case 4: fold_tile<2, 0>(lds, lane, v); break;
    case 5: fold_tile<2, 1>(lds, lane, v); break;
    case 6: fold_tile<3, 0>(lds, lane, v); break;
    default: fold_tile<3, 1>(lds, lane, v); break;
    }
    rt::block_sync_lds();                      // all waves have read the staged input
    clk.tick(1);
    rt::set_prio<0>();

    // ---- forward round A: lane (ya, xa), registers (yb, xb)
    class_twiddle<false>(tw, cy, cx, v);
    fft_8x4<1>(v);                             // -> (kyA, kxA)
    clk.tick(2);
    {
        const int ya = l5 >> 3, xa = l5 & 7;
#pragma unroll
        for (int ka = 0; ka < 8; ++ka) {
            const cf wy = tw[ya * (cy + 4 * ka)];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) v[4 * ka + kb] = cmul(v[4 * ka + kb], wy);
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const cf wx = tw[xa * (cx + 4 * kb)];
#pragma unroll
            for (int ka = 0; ka < 8; ++ka) v[4 * ka + kb] = cmul(v[4 * ka + kb], wx);
        }
    }
    clk.tick(3);
    transpose32(v, xch, lane);                 // -> lane (kyA, kxA), registers (ya, xa)
    clk.tick(4);
    fft_4x8<1>(v);                             // -> (kyB, kxB)
    clk.tick(5);
#pragma unroll
    for (int r = 0; r < 32; ++r) v[r] = cmul(v[r], v[r]);      // W = Z^2
    clk.tick(6);
    // ---- inverse round A': registers (kyB, kxB) -> (ya, xa); lane (kyA, kxA)
    fft_4x8<-1>(v);
    {
        const int ka = l5 >> 2, kb = l5 & 3;
#pragma unroll
        for (int ya = 1; ya < 4; ++ya) {
            const cf wy = tw[ya * (cy + 4 * ka)];
#pragma unroll
            for (int xa = 0; xa < 8; ++xa) v[8 * ya + xa] = cmulc(v[8 * ya + xa], wy);
        }
#pragma unroll
        for (int xa = 1; xa < 8; ++xa) {
            const cf wx = tw[xa * (cx + 4 * kb)];
#pragma unroll
            for (int ya = 0; ya < 4; ++ya) v[8 * ya + xa] = cmulc(v[8 * ya + xa], wx);
        }
    }
    clk.tick(7);
    transpose32(v, xch, lane);                 // -> lane (ya, xa), registers (kyA, kxA)
    clk.tick(8);
    fft_8x4<-1>(v);                            // -> (yb, xb)
    class_twiddle<true>(tw, cy, cx, v);        // E_c[ya + 4 yb][xa + 8 xb]
    clk.tick(9);
    rt::set_prio<1>();

    // ---- recombination (recombine8): what this wave keeps (`own`) and what it sends (`snd`)
    float own[32], snd[32];
    switch (role) {                            // (wave-uniform)
    case 0: recombine8<0, 0, 0>(v, lane, own, snd); break;
    case 1: recombine8<0, 0, 1>(v, lane, own, snd); break;
    case 2: recombine8<0, 1, 0>(v, lane, own, snd); break;
    case 3: recombine8<0, 1, 1>(v, lane, own, snd); break;
    case 4: recombine8<1, 0, 0>(v, lane, own, snd); break;
    case 5: recombine8<1, 0, 1>(v, lane, own, snd); break;
    case 6: recombine8<1, 1, 0>(v, lane, own, snd); break;
    default: recombine8<1, 1, 1>(v, lane, own, snd); break;
    }
    {
        f32x4* mine = reinterpret_cast<f32x4*>(xch);
#pragma unroll
        for (int r4 = 0; r4 < 8; ++r4)
            mine[r4 * 64 + lane] = f32x4{snd[4 * r4], snd[4 * r4 + 1], snd[4 * r4 + 2], snd[4 * r4 + 3]};
    }
    rt::block_sync_lds();
    {
        // the other wave of this plane: its buffer first gives what it sent, then takes this wave's rows
        float* other = reinterpret_cast<float*>(lds + L8::R_OFF + (role ^ 1) * L8::XCH_WAVE_BYTES);
        const f32x4* theirs = reinterpret_cast<const f32x4*>(other);
        float out[32];
#pragma unroll
        for (int r4 = 0; r4 < 8; ++r4) {
            const f32x4 q = theirs[r4 * 64 + lane];
#pragma unroll
            for (int k = 0; k < 4; ++k) out[4 * r4 + k] = e ? q[k] + own[4 * r4 + k] : own[4 * r4 + k] - q[k];
        }
        rt::wave_sync();                        // every lane has what it needs from that buffer
        // rows ya + 4 yb (+ 32 for e = 0) of plane (py, cxl), columns xa + 8 xb + 32 h; rows 32.. are the
        // second half of the plane = the e = 1 wave's buffer, rows 0..31 the e = 0 wave's
        const int ya = l5 >> 3, xa = l5 & 7;
#pragma unroll
        for (int yb = 0; yb < 8; ++yb)
#pragma unroll
            for (int xb = 0; xb < 4; ++xb) {
                const int row = ya + 4 * yb;                      // within this half of the plane
                other[row * L8::PS + plane_col(row, xa + 8 * xb + 32 * h)] = out[4 * yb + xb];
            }
    }
    rt::block_sync_lds();
    clk.tick(10);
}

// ---------------------------------------------------------------------------
// tail on the parity planes (pair_kernel's readers, 512 threads)
// ---------------------------------------------------------------------------
SPX_DEVICE float window_value8(const unsigned char* lds, int ny, int nx, int qy, int qx, float out_scale) {
    const float* planes = reinterpret_cast<const float*>(lds + L8::R_OFF);
    const int ly = conv_index(ny, qy), lx = conv_index(nx, qx);
    const int my = ly & 63, mx = lx & 63;
    const int sy = (ly >> 6) & 1, sx = (lx >> 6) & 1;
    float d[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) d[c] = planes[c * (L8::PLANE_BYTES / 4) + my * L8::PS + plane_col(my, mx)];
    const float fx = sx ? -1.0f : 1.0f, fy = sy ? -1.0f : 1.0f;
    const float acc = __builtin_fmaf(fy, __builtin_fmaf(fx, d[3], d[2]), __builtin_fmaf(fx, d[1], d[0]));
    return acc * out_scale;
}

SPX_DEVICE void coarse_argmax8(const unsigned char* lds, int ny, int nx, float& bv, int& bi) {
    const int tid = fresh_tid();
    const int loy = (ny - 1) / 2, lox = (nx - 1) / 2;
    const float ninf = -__builtin_inff();
    const int mx4 = (tid & 15) << 2;
    int qx[4];
    float fx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int mx = mx4 + e;
        const bool wrap = mx < lox;
        qx[e] = (nx - 1) + lox - mx - (wrap ? 64 : 0);
        fx[e] = wrap ? -1.0f : 1.0f;
    }
    float val[2][4];
    int rowbase[2];
    float m = ninf;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int my = (tid >> 4) + 32 * i;
        const bool wrap = my < loy;
        const int qy = (ny - 1) + loy - my - (wrap ? 64 : 0);
        const float fy = wrap ? -1.0f : 1.0f;
        rowbase[i] = qy * nx;
        f32x4 d[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            d[c] = *reinterpret_cast<const f32x4*>(lds + L8::R_OFF + c * L8::PLANE_BYTES +
                                                   (my * L8::PS + plane_col(my, mx4)) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = __builtin_fmaf(fx[e], d[1][e], d[0][e]);
            const float t = __builtin_fmaf(fx[e], d[3][e], d[2][e]);
            const float v = __builtin_fmaf(fy, t, u);
            val[i][e] = (qy >= 0 && qx[e] >= 0) ? v : ninf;
            m = __builtin_fmaxf(m, val[i][e]);
        }
    }
    int best = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = (val[i][e] == m) ? rowbase[i] + qx[e] : 0x7fffffff;
            best = idx < best ? idx : best;
        }
    bv = m;
    bi = best;
}

// Refine operands of one wave: plane c = role >> 1 (parity class (c >> 1, c & 1)), column tiles
// t = 2 (role & 1) + {0, 1}.  Table layout: spx_tables.h make_ktab.
template <int WB> struct FineTables8 {
    static constexpr bool kPreload = WB <= 2;
    f32x4 ky[kPreload ? WB : 1][4], kx[kPreload ? WB : 1][2];
    const f32x4* kty;
    const f32x4* ktx;
    SPX_DEVICE f32x4 y(int ab, int s4) const { return kPreload ? ky[ab][s4] : kty[ab * 64 * 4 + s4]; }
    SPX_DEVICE f32x4 x(int bb, int tt) const { return kPreload ? kx[bb][tt] : ktx[bb * 64 * 4 + tt]; }
};
template <int WB>
SPX_DEVICE void load_fine_tables8(FineTables8<WB>& ft, const float* __restrict__ ktab, int rot) {
    const int tid = fresh_tid();
    const int role = ((tid >> 6) + rot) & 7, lane = tid & 63;
    const int c = role >> 1, e = role & 1;
    const int cy = c >> 1, cx = c & 1;
    ktab = rt::launder(ktab);
    const f32x4* kty = reinterpret_cast<const f32x4*>(ktab) + ((size_t)(0 * 2 + cy) * WB * 64 + lane) * 4;
    const f32x4* ktx = reinterpret_cast<const f32x4*>(ktab) + ((size_t)(1 * 2 + cx) * WB * 64 + lane) * 4 + 2 * e;
    ft.kty = kty;
    ft.ktx = ktx;
    if constexpr (FineTables8<WB>::kPreload) {
#pragma unroll
        for (int b = 0; b < WB; ++b) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ft.ky[b][i] = kty[b * 64 * 4 + i];
#pragma unroll
            for (int i = 0; i < 2; ++i) ft.kx[b][i] = ktx[b * 64 * 4 + i];
        }
    }
}

template <int WB>
SPX_DEVICE void fine_window8(unsigned char* lds, const FineTables8<WB>& ft, int ny, int nx, int qyc,
                             int qxc, int rot) {
    constexpr int W = 16 * WB;
    const int tid = fresh_tid();
    const int role = ((tid >> 6) + rot) & 7, lane = tid & 63;
    const int c = role >> 1, e = role & 1;
    const int cy = c >> 1, cx = c & 1;
    const int lk = lane >> 4, lj = lane & 15;
    const float* plane = reinterpret_cast<const float*>(lds + L8::R_OFF + c * L8::PLANE_BYTES);
    float* fbuf = reinterpret_cast<float*>(lds + L8::FB_OFF);
    const int lyc = conv_index(ny, qyc), lxc = conv_index(nx, qxc);

    f32x4 acc[WB][2];
#pragma unroll
    for (int ab = 0; ab < WB; ++ab)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[ab][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    int col[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) col[t] = (lxc + 16 * (2 * e + t) + lj - 32) & 63;
    float afrag[16][2];
#pragma unroll
    for (int step = 0; step < 16; ++step) {
        const int row = (lyc + 4 * step + lk - 32) & 63;
#pragma unroll
        for (int t = 0; t < 2; ++t) afrag[step][t] = plane[row * L8::PS + plane_col(row, col[t])];
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        f32x4 kb[WB];
#pragma unroll
        for (int ab = 0; ab < WB; ++ab) kb[ab] = ft.y(ab, s4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int step = 4 * s4 + k;
            const int m = lyc + 4 * step + lk - 32;
            const float sgn = (cy && ((m >> 6) & 1)) ? -1.0f : 1.0f;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ab = 0; ab < WB; ++ab)
                    acc[ab][t] = rt::mfma_16x16x4(afrag[step][t], sgn * kb[ab][k], acc[ab][t]);
        }
    }
    f32x4 f[WB][WB];
#pragma unroll
    for (int bb = 0; bb < WB; ++bb)
#pragma unroll
        for (int ab = 0; ab < WB; ++ab) f[bb][ab] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        f32x4 ka[WB];
#pragma unroll
        for (int bb = 0; bb < WB; ++bb) ka[bb] = ft.x(bb, t);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = lxc + 16 * (2 * e + t) + 4 * lk + r - 32;
            const float sgn = (cx && ((m >> 6) & 1)) ? -1.0f : 1.0f;
#pragma unroll
            for (int bb = 0; bb < WB; ++bb)
#pragma unroll
                for (int ab = 0; ab < WB; ++ab)
                    f[bb][ab] = rt::mfma_16x16x4(sgn * ka[bb][r], acc[ab][t][r], f[bb][ab]);
        }
    }
    const float scale = 0.5f / (float)(L8::P * L8::P);
    if constexpr (L8::fb_count(W) == kW8) {
        float* mine = fbuf + role * W * W;
#pragma unroll
        for (int bb = 0; bb < WB; ++bb)
#pragma unroll
            for (int ab = 0; ab < WB; ++ab)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    mine[(bb * 16 + 4 * lk + r) * W + ab * 16 + lj] = f[bb][ab][r] * scale;
        rt::block_sync_lds();
    } else {
        for (int k = 0; k < kW8; ++k) {              // fixed order: role 0, 1, ... 7
            if (role == k) {
#pragma unroll
                for (int bb = 0; bb < WB; ++bb)
#pragma unroll
                    for (int ab = 0; ab < WB; ++ab)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int b = bb * 16 + 4 * lk + r, a = ab * 16 + lj;
                            const float val = f[bb][ab][r] * scale;
                            if (k == 0) fbuf[b * W + a] = val; else fbuf[b * W + a] += val;
                        }
            }
            rt::block_sync_lds();
        }
    }
}

template <int W> SPX_DEVICE float fine_value8(const unsigned char* lds, int b, int a) {
    const float* fbuf = reinterpret_cast<const float*>(lds + L8::FB_OFF);
    if constexpr (L8::fb_count(W) == kW8) {
        float acc = fbuf[b * W + a];
#pragma unroll
        for (int k = 1; k < kW8; ++k) acc += fbuf[k * W * W + b * W + a];
        return acc;
    } else {
        return fbuf[b * W + a];
    }
}

SPX_DEVICE float warm_next_pair8(const float* __restrict__ ref, const float* __restrict__ img) {
    const int tid = rt::thread_id();      // 512 threads: one dword per 64 bytes of the next pair
    const float* p = (tid < 256) ? ref + tid * 16 : img + (tid - 256) * 16;
    return *p;
}

// ---------------------------------------------------------------------------
// Pair kernel, 8 waves per pair.  Same arguments, results and status codes as pair_kernel.
// ---------------------------------------------------------------------------
template <int WB, int DBG, typename TIn>
SPX_DEVICE void pair8_body(const TIn* __restrict__ ref, const TIn* __restrict__ img, int ny, int nx, int U,
                           int cc_type, const float* __restrict__ ktab, double* __restrict__ out,
                           int* __restrict__ status, unsigned char* lds, PhaseClock<DBG>& clk,
                           const TIn* __restrict__ next_ref, const TIn* __restrict__ next_img, float& warm,
                           int fit_wave, double inv_u) {
    ny = rt::launder_uniform(ny);
    nx = rt::launder_uniform(nx);
    U = rt::launder_uniform(U);
    const int tid = fresh_tid();
    unsigned char* scr = lds + L8::SCR_OFF;
    const NormStatsT<TIn> ns = norm_stats8(scr, ref, img, 1, 0, ny, nx, cc_type);
    rt::consume(warm);
    const float bal = stage_pair8<TIn>(lds, ref, img, ny, nx, ns);
    rt::block_sync_lds();                       // staged
    const float oscale = 0.5f / ((float)(L8::P * L8::P) * bal);
    clk.tick(0);
    if constexpr (DBG == 1) return;
    const int rot = (8 - fit_wave) & 7;         // the fitting wave takes role 0 (cy = 0: no class twiddles)
    planes8<DBG>(lds, clk, rot);
    if constexpr (DBG == 10) return;
    // the refine stage's constant operands first, the L2 warm-up of the next pair second: a wait for the
    // tables (vmcnt counts in issue order) then does not also wait for the warm-up's trip to HBM
    FineTables8<(WB > 0 ? WB : 1)> ft;
    if constexpr (WB > 0) load_fine_tables8<WB>(ft, ktab, rot);
#ifndef SPX8_NO_WARM
    if constexpr (sizeof(TIn) == 4) if (next_ref) warm = warm_next_pair8(next_ref, next_img);
#endif
    float bv;
    int bi;
    coarse_argmax8(lds, ny, nx, bv, bi);
    block_argmax8(scr, bv, bi, 0);
    const bool nonfinite = bi == kNoIndex;      // see pair_body
    if (nonfinite) bi = 0;
    int qyc = bi / nx, qxc = bi - (bi / nx) * nx;
    clk.tick(11);
    if constexpr (DBG == 11) { if (tid == 0) out[0] = (double)bi; return; }

    PeakResult pk;
    if (nonfinite) {
        pk.x = 0.0; pk.y = 0.0; pk.status = ST_NONFINITE;
    } else if constexpr (WB == 0) {
        pk = peak_fit_wave0(scr, qxc, qyc, nx, ny, [&](int x, int y) {
            return window_value8(lds, ny, nx, y, x, oscale);
        }, fit_wave);
    } else {
        constexpr int W = 16 * (WB > 0 ? WB : 1);
        const int NX = U * nx, NY = U * ny;
        int imax = 0, jmax = 0;
        bool inside = false;
        for (int iter = 0; iter < 4; ++iter) {
            fine_window8<(WB > 0 ? WB : 1)>(lds, ft, ny, nx, qyc, qxc, rot);
            clk.tick(12);
            const int fx0 = U * qxc - W / 2, fy0 = U * qyc - W / 2;
            float fv = -__builtin_inff();
            int fi = 0x7fffffff;
#pragma unroll 4
            for (int i = (tid & 63); i < W * W; i += 64) {
                const int b = i / W, a = i % W;
                const int idx = a * W + b;
                const int gy = fy0 + a, gx = fx0 + b;
                const float val = fine_value8<W>(lds, b, a);
                const bool in = gy >= 0 && gy < NY && gx >= 0 && gx < NX;
                if (in && better(val, idx, fv, fi)) { fv = val; fi = idx; }
            }
            wave_argmax(fv, fi);
            clk.tick(13);
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
        clk.tick(16);
        if (inside) {
            const int fx0 = U * qxc - W / 2, fy0 = U * qyc - W / 2;
            pk = peak_fit_wave0(scr, imax, jmax, NX, NY, [&](int x, int y) {
                return fine_value8<W>(lds, x - fx0, y - fy0);
            }, fit_wave);
        } else if (imax < 0) {
            pk.x = 0.0; pk.y = 0.0; pk.status = ST_NONFINITE;
        } else {
            pk.x = (double)imax; pk.y = (double)jmax; pk.status = ST_WINDOW;
        }
    }
    clk.tick(17);
    if (tid == 64 * fit_wave) {
        out[0] = pk.x * inv_u - (double)((nx - 1) / 2);       // cc.py:89-93 with 2 -> U
        out[1] = pk.y * inv_u - (double)((ny - 1) / 2);
        if (status) status[0] = pk.status;
    }
    clk.tick(14);
}

template <int WB, int DBG = 0, typename TIn = float>
SPX_TKERNEL8(512) void pair8_kernel(const TIn* __restrict__ ref, const TIn* __restrict__ img,
                                    int64_t nbatch, int ny, int nx, int U, int cc_type,
                                    const cf* __restrict__ tw_g, const float* __restrict__ ktab,
                                    double* __restrict__ out, int* __restrict__ status) {
    SPX_DYN_LDS(lds);
    {
        cf* tw = reinterpret_cast<cf*>(lds + L8::TW_OFF);
        for (int i = rt::thread_id(); i < L8::P; i += kT8) tw[i] = tw_g[i];
        rt::block_sync_lds();
    }
    PhaseClock<DBG> clk;
    clk.start();
    const int64_t stride = (int64_t)ny * nx;
    const bool full = sizeof(TIn) == 4 && ny == 64 && nx == 64 &&
        ((reinterpret_cast<uintptr_t>(ref) | reinterpret_cast<uintptr_t>(img)) & 15) == 0;
    const int64_t step = rt::grid_size();
    float warm = 0.0f;
    int fit_wave = (int)(rt::block_id() & 7);
    const double inv_u = rt::read_lane(1.0 / (double)U, 0);
    for (int64_t p = first_item(rt::block_id(), step); p < nbatch; p += step) {
        const bool more = full && (p + step < nbatch);
        pair8_body<WB, DBG, TIn>(ref + p * stride, img + p * stride, ny, nx, U, cc_type, ktab,
                                 out + 2 * p, status ? status + p : nullptr, lds, clk,
                                 more ? ref + (p + step) * stride : nullptr,
                                 more ? img + (p + step) * stride : nullptr, warm, fit_wave, inv_u);
        fit_wave = (fit_wave + 1) & 7;
        // upsample > 1: after the last barrier of a pair every wave only reads the fine windows, which the
        // next pair's staging does not touch (its first barrier comes before it writes the staged input);
        // upsample = 1 fits on the planes, which the staging overwrites
        if constexpr (WB == 0) rt::block_sync_lds();
        clk.tick(15);
    }
    if constexpr (DBG == 100)
        clk.flush(reinterpret_cast<unsigned long long*>(status + nbatch));
}

}  // namespace w8
}  // namespace spx
