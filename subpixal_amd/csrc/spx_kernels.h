// spx_kernels.h -- CDNA4 (gfx950) kernels for subpixal's hot path:
// per-cutout FFT cross-correlation + sub-pixel peak refinement.
//
// Replaces, for a whole batch at once, the body of the per-source loop at
// /root/reference/subpixal/align.py:656-699, i.e.
//   cc.find_displacement   (cc.py:21-95)   -> find_displacement5 kernel
//   cc._build_icc_image    (cc.py:98-128)  -> cc_planes + icc interlace
//   cc._normalize          (cc.py:131-156) -> stage_* (in-LDS statistics)
//   centroid.find_peak     (centroid.py:18-236, hot-path arguments of cc.py:86)
//                                          -> arg-max reduction + quad_fit
// plus the pair / upsample=U mode of BASELINE.json.
//
// Design (DESIGN.md has the long form).  One 256-thread workgroup per cutout
// pair, tile T = 64 (cutouts up to 64x64, zero padded), FFT period P = 2T = 128
// as in scipy's fftconvolve (next_fast_len(2n-1)).  Zero padding makes the first
// radix-2 DIF stage free:  Z[2k'+c] = FFT64{ z[x] w_P^(c x) }[k'],  so the
// 128x128 spectrum splits into 4 parity classes (cy,cx), each a 64x64 complex
// FFT.  Wave w owns class (w>>1, w&1) and keeps its 64x64 complex points in
// registers as an 8x8 tile per lane; a 2-D FFT is two register rounds
// (radix-8 in y and x) with ONE lane<->register transposition through LDS in
// between.  The two real images are packed as z = ref + i*flip(img) (flip = both
// axes reversed, exactly the operand of the reference's fftconvolve(ref,
// im[::-1, ::-1]), cc.py:114): with Z = FFT(z),
//   IFFT(Z^2) = (ref*ref - fimg*fimg) + 2i (ref * fimg)       (* = convolution)
// so the cross-correlation is Im(IFFT(Z^2))/2 and the spectral product is a plain
// element-wise complex SQUARE -- no Z[k] / Z[-k] unpacking, no data exchange.  The
// inverse runs the same rounds backwards and leaves, per class, the REAL plane
//   d_c[l] = Im sum_{k in class c} Z[k]^2 e^{+2 pi i k l / P},   l in [0,64)^2  (twice the
//   class's share of the convolution; the readers scale by 1/(2 P^2)),
// in LDS (real because each class is closed under k -> -k).  The full linear
// convolution at index l in [0,127)^2 (lag = l - (n-1)) is
//   conv[l] = (2 P^2)^-1 sum_c (-1)^(c . [l>=64]) d_c[l mod 64]
// and its trigonometric interpolant (the upsample=U mode) is
//   F(t)  = (2 P^2)^-1 sum_c sum_m K_cy(ty-my) d_c[m] K_cx(tx-mx),
// two small real matrix products per class, done with v_mfma_f32_16x16x4_f32.
//
// This header is compiled by hipcc for gfx950 (spx_capi.hip) and, unchanged, by
// the CPU logic-check harness of the unit tests (tests/cpu_emu).  It needs the
// names of spx_rt_hip.h (or the harness's equivalent) to be declared first.
#pragma once

namespace spx {

using rt::f32x2;
using rt::f32x4;
typedef f32x2 cf;   // complex float: .x = re, .y = im

constexpr int kThreads = 256;

// Work-item id made opaque to the optimiser: values derived from it (LDS addresses,
// lane masks) are then recomputed in the phase that needs them instead of being
// hoisted out of the per-pair loop and kept -- i.e. spilled -- for the whole kernel.
SPX_DEVICE int fresh_tid() { return rt::launder_lane(rt::thread_id()); }

enum { ST_OK = 0, ST_EDGE = 1, ST_NOMAX = 2, ST_OUTSIDE = 3, ST_WINDOW = 4, ST_FEWPTS = 5, ST_NONFINITE = 6, ST_SHAPE = 7 };
// index of an arg-max that saw no comparable value (NaN everywhere)
constexpr int kNoIndex = 0x7fffffff;
enum { CC_PLAIN = 0, CC_NCC = 1, CC_ZNCC = 2 };

// ---------------------------------------------------------------------------
// Per-phase cycle accounting of the diagnostic build (DBG == 100, `make diag`):
// every wave stamps the shader clock at phase boundaries and adds its totals to a
// global array at the end.  With any other DBG the methods compile to nothing.
// ---------------------------------------------------------------------------
constexpr int kNumPhases = 20;
template <int DBG> struct PhaseClock {
    unsigned long long last;
    unsigned long long acc[DBG == 100 ? kNumPhases : 1];
    SPX_DEVICE void start() {
        if constexpr (DBG == 100) {
            for (int i = 0; i < kNumPhases; ++i) acc[i] = 0;
            last = rt::clock_stamp();
        }
    }
    SPX_DEVICE void tick(int phase) {
        if constexpr (DBG == 100) {
            const unsigned long long now = rt::clock_stamp();
            acc[phase] += now - last;
            last = now;
        }
    }
    SPX_DEVICE void flush(unsigned long long* out) {
        if constexpr (DBG == 100) {
            if ((rt::thread_id() & 63) == 0)
                for (int i = 0; i < kNumPhases; ++i) rt::atomic_add_u64(out + i, acc[i]);
        }
    }
};

// ---------------------------------------------------------------------------
// complex helpers.  The primitives (cmul, cmulc, add_mi, add_pi, ...) are single
// packed VOP3P instructions on gfx950 (spx_rt_hip.h).
// ---------------------------------------------------------------------------
SPX_DEVICE cf cmul(cf a, cf w) { return rt::cmul(a, w); }
SPX_DEVICE cf cmulc(cf a, cf w) { return rt::cmulc(a, w); }
// s + rot(d) and s - rot(d), rot = multiply by -i (forward, DIR > 0) or +i (inverse)
template <int DIR> SPX_DEVICE cf rot_add(cf s, cf d) { return DIR > 0 ? rt::add_mi(s, d) : rt::add_pi(s, d); }
template <int DIR> SPX_DEVICE cf rot_sub(cf s, cf d) { return DIR > 0 ? rt::add_pi(s, d) : rt::add_mi(s, d); }

// 4-point DFT, natural order in and out; `c2r` is c2 BEFORE its rotation by -+i when
// ROT2 is set (lets the radix-8 stage hand over (a2 - a6) unrotated)
template <int DIR, bool ROT2>
SPX_DEVICE void fft4(cf c0, cf c1, cf c2, cf c3, cf& y0, cf& y1, cf& y2, cf& y3) {
    const cf s0 = ROT2 ? rot_add<DIR>(c0, c2) : c0 + c2;
    const cf s1 = ROT2 ? rot_sub<DIR>(c0, c2) : c0 - c2;
    const cf s2 = c1 + c3, d = c1 - c3;
    y0 = s0 + s2;
    y2 = s0 - s2;
    y1 = rot_add<DIR>(s1, d);
    y3 = rot_sub<DIR>(s1, d);
}

// 8-point DFT, natural order in and out (radix-2 DIF + two radix-4):
// X[k] = sum_j a[j] e^{-DIR 2 pi i j k / 8}
// (Round 3 tried folding the two multiplications by h into fused multiply-adds of the last stage -- 26 packed
// instructions instead of 28, SQ_INSTS_VALU 4130 -> 4074 per wave-pair: no gain on the four-wave kernel, 5 % slower
// on the eight-wave one, profiles/r03/variants_pt_ftfirst_w8_nowarm.txt; the constants then ride in scalar
// register pairs of every v_pk_fma.  Dropped.)
template <int DIR> SPX_DEVICE void fft8(cf (&a)[8]) {
    const float h = 0.70710678118654752440f;
    const cf b0 = a[0] + a[4], d4 = a[0] - a[4];
    const cf b1 = a[1] + a[5], d5 = a[1] - a[5];
    const cf b2 = a[2] + a[6], d6 = a[2] - a[6];
    const cf b3 = a[3] + a[7], d7 = a[3] - a[7];
    // w8^1 d5 = h (d5 + rot d5),  w8^3 d7 = h (-d7 + rot d7)
    const cf b5 = rot_add<DIR>(d5, d5) * h;
    const cf b7 = (DIR > 0 ? rt::neg_add_mi(d7) : rt::neg_add_pi(d7)) * h;
    fft4<DIR, false>(b0, b1, b2, b3, a[0], a[2], a[4], a[6]);
    fft4<DIR, true>(d4, b5, d6, b7, a[1], a[3], a[5], a[7]);
}

// radix-8 along the first (y) digit of an 8x8 register tile
template <int DIR> SPX_DEVICE void fft8_y(cf (&v)[8][8]) {
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        cf t[8];
#pragma unroll
        for (int y = 0; y < 8; ++y) t[y] = v[y][x];
        fft8<DIR>(t);
#pragma unroll
        for (int y = 0; y < 8; ++y) v[y][x] = t[y];
    }
}
// radix-8 along the second (x) digit
template <int DIR> SPX_DEVICE void fft8_x(cf (&v)[8][8]) {
#pragma unroll
    for (int y = 0; y < 8; ++y) fft8<DIR>(v[y]);
}

// one row / one column of the tile times w (CONJ: times conj(w)), in place
#ifndef SPX_CMUL_BATCH
#define SPX_CMUL_BATCH 1
#endif
template <bool CONJ> SPX_DEVICE void mul_row(cf (&v)[8][8], int y, cf w) {
    if constexpr (SPX_CMUL_BATCH) {
        rt::cmul8_ip<CONJ>(v[y][0], v[y][1], v[y][2], v[y][3], v[y][4], v[y][5], v[y][6], v[y][7], w);
    } else {
#pragma unroll
        for (int x = 0; x < 8; ++x) { if (CONJ) rt::cmulc_ip(v[y][x], w); else rt::cmul_ip(v[y][x], w); }
    }
}
template <bool CONJ> SPX_DEVICE void mul_col(cf (&v)[8][8], int x, cf w) {
    if constexpr (SPX_CMUL_BATCH) {
        rt::cmul8_ip<CONJ>(v[0][x], v[1][x], v[2][x], v[3][x], v[4][x], v[5][x], v[6][x], v[7][x], w);
    } else {
#pragma unroll
        for (int y = 0; y < 8; ++y) { if (CONJ) rt::cmulc_ip(v[y][x], w); else rt::cmul_ip(v[y][x], w); }
    }
}

// lane <-> register transposition of an 8x8 complex tile through the wave's own LDS
// buffer (64 rows of XS floats): (lane L, register R) -> (lane R, register L); real and
// imaginary parts in two passes (ds_write2_b32 / ds_read_b128, conflict-free at XS = 68).
template <int XS> SPX_DEVICE void transpose_tile(cf (&v)[8][8], float* xch, int lane) {
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int r = 0; r < 64; ++r)
            xch[r * XS + lane] = part ? v[r >> 3][r & 7].y : v[r >> 3][r & 7].x;
        rt::wave_sync();
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            const float t = xch[lane * XS + r];
            if (part) v[r >> 3][r & 7].y = t; else v[r >> 3][r & 7].x = t;
        }
        rt::wave_sync();
    }
}

// The same two passes with 4-byte reads (rt::lds_read_f32: never merged into ds_read_b128): a 16-byte
// read lands in four consecutive registers, i.e. in the .x / .y slots of TWO elements, and every value
// then needs a v_mov into its own (re, im) pair -- 64 vector moves per pass on the unit that bounds
// the kernel; 4-byte reads land where they are used and cost LDS issue slots instead.
template <int XS> SPX_DEVICE void transpose_tile_v(cf (&v)[8][8], float* xch, int lane) {
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int r = 0; r < 64; ++r)
            xch[r * XS + lane] = part ? v[r >> 3][r & 7].y : v[r >> 3][r & 7].x;
        rt::wave_sync();
#pragma unroll
        for (int r = 0; r < 64; ++r) {
            const float t = rt::lds_read_f32(xch + lane * XS + r);
            if (part) v[r >> 3][r & 7].y = t; else v[r >> 3][r & 7].x = t;
        }
        rt::wave_sync();
    }
}

// Whole complex elements with the passes split by SOURCE lane half: in pass h the lanes of half h write
// their 64 registers as (re, im) slots (row R, slot lane & 31: ds_write_b64, 256 contiguous bytes per
// row), then ALL lanes read the 32 slots of their row with 16 ds_read_b128 -- two elements per read,
// in place.  Half-wave writes instead of transpose_tile_cplx's half-wave reads, no moves either.
template <int XS> SPX_DEVICE void transpose_tile_w(cf (&v)[8][8], float* xch, int lane) {
    static_assert(XS >= 68 && (XS & 3) == 0, "32 complex slots per row plus the conflict padding");
    cf* buf = reinterpret_cast<cf*>(xch);
    cf t[64];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if ((lane >> 5) == h) {
#pragma unroll
            for (int r = 0; r < 64; ++r) buf[r * (XS / 2) + (lane & 31)] = v[r >> 3][r & 7];
        }
        rt::wave_sync();
        const f32x4* row = reinterpret_cast<const f32x4*>(buf + lane * (XS / 2));
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const f32x4 q = row[j];
            t[32 * h + 2 * j] = cf{q[0], q[1]};
            t[32 * h + 2 * j + 1] = cf{q[2], q[3]};
        }
        rt::wave_sync();
    }
#pragma unroll
    for (int r = 0; r < 64; ++r) v[r >> 3][r & 7] = t[r];
}

#ifndef SPX_TRANSPOSE_VARIANT
#define SPX_TRANSPOSE_VARIANT 0
#endif
template <int XS> SPX_DEVICE void transpose_tile_sel(cf (&v)[8][8], float* xch, int lane) {
    if constexpr (SPX_TRANSPOSE_VARIANT == 1) transpose_tile_v<XS>(v, xch, lane);
    else if constexpr (SPX_TRANSPOSE_VARIANT == 2) transpose_tile_w<XS>(v, xch, lane);
    else transpose_tile<XS>(v, xch, lane);
}

// A class plane on its way from the tile registers to 16-byte row-contiguous stores goes through the
// wave's exchange buffer as 64 rows of 64 floats.  Lane (l1, l0) writes rows l1 + 8 y1, columns
// l0 + 8 x1: with a plain row stride of 64 floats all eight l1 of a column share one bank (8-way
// conflict on every ds_write_b32: SQ_LDS_BANK_CONFLICT was 92 % of the period-192 kernel's LDS
// cycles).  XOR-ing the 8-column group with the row's low three bits spreads them over all 64
// banks and keeps every aligned group of 4 columns contiguous, so the 16-byte reads stay whole:
//   float element (row, col)      -> row * 64 + (col ^ ((row & 7) << 3))
//   float4 slot n = row * 16 + c4 -> n ^ (((n >> 4) & 7) << 1)
SPX_DEVICE int plane_elem(int row, int col) { return row * 64 + (col ^ ((row & 7) << 3)); }
SPX_DEVICE int plane_slot(int n) { return n ^ (((n >> 4) & 7) << 1); }

// The same transposition with whole complex elements (8 bytes) in two half-tile passes: pass h
// moves the registers R in [32 h, 32 h + 32) as 32 rows of 66 complex slots (ds_write_b64 by all
// lanes, conflict-free), and the lanes R of that range read their row back with 32 ds_read_b128.
// No register shuffling on either side (-368 VALU instructions, -19 VGPRs) but the reads run on
// half a wave: measured neutral to slightly slower on the 64 / 32 tiles, so it is used where the
// registers matter more than the LDS issue slots (period-192 and general paths, 32-tile reference
// mode: their spills go away).
template <int XS> SPX_DEVICE void transpose_tile_cplx(cf (&v)[8][8], float* xch, int lane) {
    constexpr int RS = 66;                                   // row stride in complex slots
    static_assert(32 * RS * 2 <= 64 * XS, "half a tile must fit the wave's exchange buffer");
    cf* buf = reinterpret_cast<cf*>(xch);
    cf t[64];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int r = 0; r < 32; ++r) buf[r * RS + lane] = v[(32 * h + r) >> 3][(32 * h + r) & 7];
        rt::wave_sync();
        if ((lane >> 5) == h) {
            const f32x4* row = reinterpret_cast<const f32x4*>(buf + (lane & 31) * RS);
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const f32x4 q = row[j];
                t[2 * j] = cf{q[0], q[1]};
                t[2 * j + 1] = cf{q[2], q[3]};
            }
        }
        rt::wave_sync();
    }
#pragma unroll
    for (int r = 0; r < 64; ++r) v[r >> 3][r & 7] = t[r];
}

// ---------------------------------------------------------------------------
// LDS map (bytes).  Everything is carved from one dynamic region.
// ---------------------------------------------------------------------------
template <int C> struct Lds {
    static constexpr int P = 64 * C;          // FFT period
    static constexpr int T = 32 * C;          // tile (max cutout side)
    static constexpr int NCLS = C * C;        // parity classes = waves
    static constexpr int ZS = 72;             // staged-input row stride (floats)
    static constexpr int XS = 68;             // transposition row stride (floats)
    static constexpr int PS = 64;             // class-plane row stride (floats)
    static constexpr int TW_OFF = 0;                      // cf[P]
    static constexpr int SCR_OFF = TW_OFF + P * 8;        // 1 KiB scratch
    static constexpr int R_OFF = SCR_OFF + 1024;          // phase-shared region
    static constexpr int ZBUF_BYTES = 2 * 64 * ZS * 4;
    static constexpr int XCH_WAVE_BYTES = 64 * XS * 4;
    static constexpr int XCH_BYTES = NCLS * XCH_WAVE_BYTES;
    // class plane c occupies the head of wave c's OWN exchange buffer, so no other
    // wave's exchange traffic can touch it (no barrier needed before writing it)
    static constexpr int PLANE_STRIDE_BYTES = XCH_WAVE_BYTES;
    static constexpr int FB_OFF = R_OFF + XCH_BYTES;        // fine windows
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    // fine window(s) of W x W floats sit behind the exchange region: one per class for
    // W = 16 (summed by the reader, one barrier), a single shared one above that (summed by the
    // writers in class order, four barriers) so that two workgroups still fit a CU's 160 KiB
    // up to W = 48 (75.6 / 75.6 / 80.9 KiB; W = 64: 88 KiB, one workgroup per CU)
    static constexpr int fb_count(int W) { return W <= 16 ? NCLS : 1; }
    static constexpr int total(int W) {
        return cmax(R_OFF + cmax(ZBUF_BYTES, XCH_BYTES), FB_OFF + fb_count(W) * W * W * 4);
    }
};

// scratch (1 KiB) sub-offsets
constexpr int SCR_RED_F = 0;      // 2 slots x (float[4] values + int[4] indices)
constexpr int SCR_INT = 64;       // int[16]   broadcast integers
constexpr int SCR_RED_D = 128;    // double[4*4] per-wave double partials
constexpr int SCR_FIT = 256;      // double[25] fit box values
constexpr int SCR_STAT = 512;     // double[8]  statistics
constexpr int SCR_RED_D6 = 576;   // double[4*6] per-wave partials of the six-value reduction (norm_stats)

// ---------------------------------------------------------------------------
// workgroup reductions
// ---------------------------------------------------------------------------
SPX_DEVICE double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += rt::shfl_xor(v, m);
    return v;
}

// sums NV doubles over the workgroup (fixed order: lanes by butterfly, waves
// 0..3); every thread gets the totals.  Two barriers.
template <int NV> SPX_DEVICE void block_sum(unsigned char* lds_scr, double (&v)[NV]) {
    static_assert(NV <= 4 || NV == 6, "scratch slots exist for up to 4 values, or exactly 6");
    constexpr int ST = NV <= 4 ? 4 : 6;
    const int tid = rt::thread_id();
    double* part = reinterpret_cast<double*>(lds_scr + (NV <= 4 ? SCR_RED_D : SCR_RED_D6));
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) part[(tid >> 6) * ST + i] = v[i];
    }
    rt::block_sync_lds();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double s = 0.0;
        for (int w = 0; w < kThreads / 64; ++w) s += part[w * ST + i];
        v[i] = s;
    }
    rt::block_sync_lds();
}

// float version for quantities that only steer scaling (sum of squares): one barrier.
// The scratch slots are next written after at least one more workgroup barrier.
SPX_DEVICE void block_sum2f(unsigned char* lds_scr, float& a, float& b) {
    const int tid = rt::thread_id();
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        a += rt::shfl_xor(a, m);
        b += rt::shfl_xor(b, m);
    }
    float* part = reinterpret_cast<float*>(lds_scr + SCR_STAT);
    if ((tid & 63) == 0) { part[2 * (tid >> 6)] = a; part[2 * (tid >> 6) + 1] = b; }
    rt::block_sync_lds();
    a = (part[0] + part[2]) + (part[4] + part[6]);
    b = (part[1] + part[3]) + (part[5] + part[7]);
}

// (value, index) arg-max with centroid.py:114's tie rule: the FIRST maximum in
// row-major order, i.e. larger value wins, equal values -> smaller index.
SPX_DEVICE bool better(float v, int i, float bv, int bi) {
    return (v > bv) || (v == bv && i < bi);
}
// Reference mode: numpy.argmax treats NaN as the maximum and returns the FIRST one
// (centroid.py:114).  A NaN is ranked as +inf here, so the (value, lowest index) reductions
// find that position; a best value of +inf afterwards means "non-finite correlation".
SPX_DEVICE float nan_as_inf(float v) { return v != v ? __builtin_inff() : v; }

// `slot` (0/1) selects one of two scratch areas: consecutive calls alternate slots, and a
// slot is only rewritten after at least one more workgroup barrier, so no trailing
// barrier is needed.
template <int STEP> SPX_DEVICE void argmax_step(float& v, int& idx) {
    const float ov = rt::row_xchg<STEP>(v);
    const int oi = rt::row_xchg<STEP>(idx);
    if (better(ov, oi, v, idx)) { v = ov; idx = oi; }
}
// wave-wide (value, index) arg-max: 4 DPP steps inside each row of 16 lanes, then the
// four row results through readlane (scalar)
SPX_DEVICE void wave_argmax(float& v, int& idx) {
    argmax_step<0>(v, idx);
    argmax_step<1>(v, idx);
    argmax_step<2>(v, idx);
    argmax_step<3>(v, idx);
    float bv = rt::read_lane(v, 0);
    int bi = rt::read_lane(idx, 0);
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        const float ov = rt::read_lane(v, 16 * r);
        const int oi = rt::read_lane(idx, 16 * r);
        if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    v = bv;
    idx = bi;
}

SPX_DEVICE void block_argmax(unsigned char* lds_scr, float& v, int& idx, int slot) {
    const int tid = rt::thread_id();
    wave_argmax(v, idx);
    float* rf = reinterpret_cast<float*>(lds_scr + SCR_RED_F) + 8 * slot;
    int* ri = reinterpret_cast<int*>(lds_scr + SCR_RED_F) + 8 * slot + 4;
    if ((tid & 63) == 0) { rf[tid >> 6] = v; ri[tid >> 6] = idx; }
    rt::block_sync_lds();
    v = rf[0];
    idx = ri[0];
    for (int w = 1; w < kThreads / 64; ++w)
        if (better(rf[w], ri[w], v, idx)) { v = rf[w]; idx = ri[w]; }
}

// ---------------------------------------------------------------------------
// 5x5 quadratic least squares as a constant operator:
// c = pinv(V) d on box-relative, centred coordinates (-2..2)^2, float64.
// Rows: 1, x, y, xy, x^2, y^2.  Replaces numpy.linalg.lstsq at centroid.py:207.
// For the centred 5x5 grid the normal equations decouple:
//   c10 = sum(x d)/50, c01 = sum(y d)/50, c11 = sum(xy d)/100,
//   c20 = (sum(x^2 d) - 2 sum(d))/70, c02 = (sum(y^2 d) - 2 sum(d))/70.
// ---------------------------------------------------------------------------
struct PeakResult {
    double x, y;
    int status;
};

// the fit from the six moment sums of the box whose first pixel is (x1, y1);
// (imax, jmax): integer arg-max; (nx, ny): image size (centroid.py:217-236).
SPX_DEVICE PeakResult quad_fit_finish(double s0, double sx, double sy, double sxy, double sxx,
                                      double syy, int x1, int y1, int imax, int jmax, int nx,
                                      int ny) {
    const double c10 = sx * (1.0 / 50.0), c01 = sy * (1.0 / 50.0), c11 = sxy * (1.0 / 100.0);
    const double c20 = (sxx - 2.0 * s0) * (1.0 / 70.0), c02 = (syy - 2.0 * s0) * (1.0 / 70.0);
    PeakResult r;
    const double det = 4.0 * c02 * c20 - c11 * c11;
    if (det <= 0.0 || ((c20 > 0.0 && c02 >= 0.0) || (c20 >= 0.0 && c02 > 0.0))) {
        r.x = x1 + 2.5;           // (x1 + x2)/2 with x2 exclusive: centroid.py:225
        r.y = y1 + 2.5;
        r.status = ST_NOMAX;
        return r;
    }
    const double inv_det = 1.0 / det;
    const double xm = (x1 + 2) + (c01 * c11 - 2.0 * c02 * c10) * inv_det;
    const double ym = (y1 + 2) + (c10 * c11 - 2.0 * c01 * c20) * inv_det;
    if (xm > 0.0 && xm < nx - 1.0 && ym > 0.0 && ym < ny - 1.0) {
        r.x = xm;
        r.y = ym;
        r.status = ST_OK;
    } else {
        r.x = (double)imax;       // centroid.py:230-236 (auto_expand_search False)
        r.y = (double)jmax;
        r.status = ST_OUTSIDE;
    }
    return r;
}

// d: 25 values, row-major (y outer), of the box whose first pixel is (x1, y1)
SPX_DEVICE PeakResult quad_fit_5x5(const double* d, int x1, int y1, int imax, int jmax,
                                   int nx, int ny) {
    double s0 = 0, sx = 0, sy = 0, sxy = 0, sxx = 0, syy = 0;
    for (int j = 0; j < 5; ++j) {
        for (int i = 0; i < 5; ++i) {
            double v = d[j * 5 + i];
            double x = (double)(i - 2), y = (double)(j - 2);
            s0 += v;
            sx += x * v;
            sy += y * v;
            sxy += x * y * v;
            sxx += x * x * v;
            syy += y * y * v;
        }
    }
    return quad_fit_finish(s0, sx, sy, sxy, sxx, syy, x1, y1, imax, jmax, nx, ny);
}

// The same fit by one whole wave, without LDS: lanes 0..24 pass their box value `v`
// (row-major); lane j < 6 forms moment j = sum_y wy(y) sum_x wx(x) v(x, y) with
// (wx, wy) = (1,1) (x,1) (1,y) (x,y) (x^2,1) (1,y^2); every lane then finishes the
// arithmetic on the broadcast sums.  All 64 lanes must call it.
SPX_DEVICE PeakResult quad_fit_wave(float v, int lane, int x1, int y1, int imax, int jmax,
                                    int nx, int ny) {
    lane = rt::launder_lane(lane);                // keeps the weights out of long-lived registers
    const int j = lane < 6 ? lane : 0;
    const int px = (j == 1 || j == 3) ? 1 : (j == 4 ? 2 : 0);
    const int py = (j == 2 || j == 3) ? 1 : (j == 5 ? 2 : 0);
    double xw[5], yw[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const double c = (double)(i - 2);
        xw[i] = px == 0 ? 1.0 : (px == 1 ? c : c * c);
        yw[i] = py == 0 ? 1.0 : (py == 1 ? c : c * c);
    }
    double s = 0.0;
#pragma unroll
    for (int y = 0; y < 5; ++y) {
        double row = 0.0;
#pragma unroll
        for (int x = 0; x < 5; ++x) row += xw[x] * (double)rt::read_lane(v, 5 * y + x);
        s += yw[y] * row;
    }
    const double s0 = rt::read_lane(s, 0), sx = rt::read_lane(s, 1), sy = rt::read_lane(s, 2);
    const double sxy = rt::read_lane(s, 3), sxx = rt::read_lane(s, 4), syy = rt::read_lane(s, 5);
    return quad_fit_finish(s0, sx, sy, sxy, sxx, syy, x1, y1, imax, jmax, nx, ny);
}

// First item of workgroup b's grid-stride walk over the batch (item = b' + k * grid).  Workgroups are
// dealt round-robin over the 8 XCDs (observed placement, MI355X_MICROARCH.md "Workgroup dispatch":
// blocks b and b + 8 share an L2; a speed assumption only).  b' hands each XCD RUNS OF 8 CONSECUTIVE
// ITEMS: their 16-byte shift records fill one 128-byte line and their status words one 32-byte sector
// in that XCD's L2 and leave it whole, instead of eight L2s each writing a fragment (the 3.1x write
// amplification of round 1: now 1.000x).  The set of items in flight at any time is the same as for
// b' = b, so the read side does not notice.  (Giving each XCD one long contiguous run instead
// costs 5-9 %: eight windows a power of two apart land on the same memory channels.)  Launches that are
// not a multiple of 64 workgroups (small batches) walk linearly.
SPX_DEVICE int64_t first_item(int64_t b, int64_t nwg) {
#ifdef SPX_LINEAR_WALK          // A/B measurements only (tools/gpu_walk_ab.sh)
    return b;
#endif
    if (nwg & 63) return b;
    const int64_t x = b & 7, j = b >> 3;
    return ((j >> 3) << 6) + (x << 3) + (j & 7);
}

// the w_P^j table lives in LDS for the whole life of the workgroup
template <int C> SPX_DEVICE void load_twiddles(unsigned char* lds, const cf* __restrict__ tw_g) {
    typedef Lds<C> L;
    cf* tw = reinterpret_cast<cf*>(lds + L::TW_OFF);
    for (int i = rt::thread_id(); i < L::P; i += kThreads) tw[i] = tw_g[i];
    rt::block_sync_lds();
}

// L2 warm-up of the NEXT pair, issued when the current pair enters its arg-max /
// refine / fit tail: one dword per 128-byte line (256 threads x 128 B = the 32 KiB of a
// full 64x64 pair).  The value is only kept alive until the next staging, where the
// real 16-byte loads then hit L2 instead of HBM.  (Holding the whole next pair in
// registers does not work: hipcc spills them to scratch right after the loads, which
// stalls the tail on HBM latency -- measured 20.3e6 vs 22.0e6 pairs/s.)
SPX_DEVICE float warm_next_pair(const float* __restrict__ ref, const float* __restrict__ img) {
    const int tid = rt::thread_id();
    const float* p = (tid < 128) ? ref + tid * 32 : img + (tid - 128) * 32;
    return *p;
}

// ---------------------------------------------------------------------------
// Stage one (ref, flipped img) pair into the LDS input planes, zero padded to 64x64,
// with cc.py:131-156's normalisation (pool = this one image, or the `npool`
// images of the 5-image mode whose statistics the caller passes in).
// ---------------------------------------------------------------------------
// what _normalize applies: im = (im - mean)/std on im != 0, in the INPUT's own type (the
// reference computes in the input dtype, cc.py:135-154: for float64 cutouts the mask, mean, std
// and the normalised pixels are float64, and only then are the pixels rounded to float32 for
// the transforms)
// (im_rstd, ref_rstd: RECIPROCALS of the standard deviations -- the reference divides every pixel by the std,
// cc.py:146-154; multiplying by the reciprocal, formed once in float64, differs from that by at most one unit in
// the last place of a float32 pixel and saves a division sequence per staged pixel: ~8 % of the reference-mode
// kernels' vector instructions for NCC / ZNCC)
template <typename T> struct NormStatsT {
    T im_mean, im_rstd, ref_mean, ref_rstd;
    int active;         // 0: plain CC
};

// Input element types: float32 (the type BASELINE.json measures) and float64 (read, masked and
// normalised as float64, then rounded to float32).  Four consecutive pixels starting at p,
// 16-byte loads that only need the element's own alignment.
template <typename T> struct Quad { T v[4]; };
struct __attribute__((packed, aligned(4))) PackedF4 { float v[4]; };
struct __attribute__((packed, aligned(8))) PackedD2 { double v[2]; };
struct __attribute__((packed, aligned(4))) PackedF2 { float v[2]; };      // 8-byte access, 4-byte aligned
SPX_DEVICE Quad<float> load_quad(const float* p) {
    const PackedF4 t = *reinterpret_cast<const PackedF4*>(p);
    return Quad<float>{{t.v[0], t.v[1], t.v[2], t.v[3]}};
}
SPX_DEVICE Quad<double> load_quad(const double* p) {
    const PackedD2 a = *reinterpret_cast<const PackedD2*>(p);
    const PackedD2 b = *reinterpret_cast<const PackedD2*>(p + 2);
    return Quad<double>{{a.v[0], a.v[1], b.v[0], b.v[1]}};
}
// one image pixel / one reference pixel as staged (cc.py:144-154)
template <typename T> SPX_DEVICE float norm_im(T m, const NormStatsT<T>& ns) {
    if (m != (T)0) { m = m - ns.im_mean; m = m * ns.im_rstd; }     // masked pixels only
    return (float)m;
}
template <typename T> SPX_DEVICE float norm_ref(T r, const NormStatsT<T>& ns) {
    r = r - ns.ref_mean;                                          // all pixels
    return (float)(r * ns.ref_rstd);
}

// ---------------------------------------------------------------------------
// Branch-free fetch of one 4-pixel chunk of the staged operand z = ref + i*flip(img): row y,
// columns x..x+3 of ref and the same positions of the flipped image (cc.py:114: pixel (y, x) of
// flip(img) is img[ny-1-y][nx-1-x]).  No control flow around the loads, so a thread's loads of
// several chunks are all in flight together (one exposed memory latency instead of one per chunk).
// Chunks outside the cutout load a clamped (valid) address and come back as zeros; a chunk that
// straddles the row end (nx not a multiple of 4) loads the row's LAST four pixels and shifts.
// That needs rows of at least four pixels.  Cutouts narrower than a chunk (3 pixels: reference
// mode's lower limit) take the NARROW instantiation -- element loads of the pixels that exist --
// chosen by a branch AROUND the staging routine (uniform per item), never per chunk: a branch per
// chunk serialises the loads, and a 4-element load of a 3-pixel row would start one element in
// front of the item (a GPU memory fault for the first item of a batch; tools/sweep_disp5.py).
// ---------------------------------------------------------------------------
template <typename TIn> struct ChunkLoad {
    Quad<TIn> r, t;      // ref[yy][xx-s .. +3],  img[ny-1-yy][nx-4-xx+s .. +3]
    int s;               // pixels of the chunk beyond the row end (0..3)
    bool in;             // chunk starts inside the cutout
};
// NX4: the caller knows nx is a multiple of 4 (uniform per item): no chunk straddles a row end, the
// shift is the constant 0 and chunk_unpack's selects fold away (a third of the staging's VALU work).
template <typename TIn, bool NARROW, bool NX4 = false>
SPX_DEVICE ChunkLoad<TIn> chunk_issue(const TIn* __restrict__ ref, const TIn* __restrict__ img,
                                      int ny, int nx, int y, int x) {
    ChunkLoad<TIn> c;
    c.in = y < ny && x < nx;
    const int yy = y < ny ? y : ny - 1;
    if (NARROW) {
        // nx < 4: same register layout as a chunk that straddles the row end by s = 4 - nx pixels
        c.s = 4 - nx;
        const TIn* rrow = ref + (int64_t)yy * nx;
        const TIn* mrow = img + (int64_t)(ny - 1 - yy) * nx;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            c.r.v[k] = k >= c.s ? rrow[k - c.s] : (TIn)0;       // r.v[s + e] = ref pixel e
            c.t.v[k] = k < nx ? mrow[k] : (TIn)0;               // t.v[3 - s - e] = flipped-image pixel e
        }
        c.in = c.in && x == 0;
        return c;
    }
    const int xx = x < nx ? x : 0;
    int s = xx + 4 - nx;
    s = (NX4 || s < 0) ? 0 : s;
    c.s = s;
    c.r = load_quad(ref + (int64_t)yy * nx + (xx - s));
    c.t = load_quad(img + (int64_t)(ny - 1 - yy) * nx + (nx - 4 - xx + s));
    return c;
}
template <typename T> SPX_DEVICE T quad_pick(const Quad<T>& q, int k) {      // q.v[k], k in [0, 4)
    const T lo = k & 1 ? q.v[1] : q.v[0], hi = k & 1 ? q.v[3] : q.v[2];
    return k & 2 ? hi : lo;
}
// (re, im)[e] = (ref, flipped img) at column x + e, normalised (cc.py:144-154); zeros outside
template <typename TIn>
SPX_DEVICE void chunk_unpack(const ChunkLoad<TIn>& c, const NormStatsT<TIn>& ns, float (&re)[4],
                             float (&im)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool ok = c.in && e + c.s < 4;
        const TIn ri = quad_pick(c.r, (e + c.s) & 3);
        const TIn mi = quad_pick(c.t, (3 - c.s - e) & 3);
        float r = (float)ri, m = (float)mi;
        if (ns.active) {
            m = norm_im(mi, ns);
            r = norm_ref(ri, ns);
        }
        re[e] = ok ? r : 0.0f;
        im[e] = ok ? m : 0.0f;
    }
}

// Staged-input geometry.  Cutouts up to 64 px: 64 rows of ZS = 72 floats.  FOLD (65..85 px,
// the period-128 transform of a cutout longer than half the period): the whole cutout plus
// zero padding in an 88 x 88 region of row stride 88 (8 consecutive columns of 8 consecutive
// rows fall into 64 distinct banks for both strides); the four parity classes then read
// z[y][x] +- z[y][x+64] +- z[y+64][x] +- z[y+64][x+64] (cc_planes).
// Cutouts up to 64 px (round 3): a row is stored PERMUTED -- pixel x = x0 + 8 x1 at float x0 * PS12 + x1,
// PS12 = 12 -- so that the eight stride-8 samples a lane's registers take from a row (round A of cc_planes:
// lane (y0, x0), registers (y1, x1)) are contiguous: two 16-byte LDS reads per row and plane instead of eight
// 4-byte ones (128 -> 32 read instructions per wave and pair).  With a row stride of 96 floats the 16-byte reads
// are bank-conflict free and the (now 4-byte) staging writes 2-way, which costs them nothing (brute-forced).
template <int C, bool FOLD> struct StageGeom {
    static constexpr int ZS = FOLD ? 88 : 96;
    static constexpr int ROWS = FOLD ? 88 : 64;
    static constexpr int CHUNKS = (FOLD ? 88 : 64) / 4;        // 4-pixel chunks per staged row
    static constexpr int PS12 = 12;                            // floats per x0 slot of a permuted row
    static_assert(2 * ROWS * ZS * 4 <= Lds<C>::XCH_BYTES, "the staged input must fit the exchange region it shares");
    // float offset inside a permuted row of pixel x = 4 q + e (q: chunk index)
    static SPX_DEVICE int perm(int q, int e) { return (4 * (q & 1) + e) * PS12 + (q >> 1); }
};

// ssq[0] += sum ref^2, ssq[1] += sum img^2 over this thread's pixels (as staged).
template <int C, bool FOLD = false, typename TIn = float, bool NARROW = false, bool NX4 = false>
SPX_DEVICE void stage_pair_rows(unsigned char* lds, const TIn* __restrict__ ref,
                           const TIn* __restrict__ img, int ny, int nx,
                           const NormStatsT<TIn>& ns, float (&ssq)[2]) {
    typedef Lds<C> L;
    typedef StageGeom<C, FOLD> G;
    float sr = 0.0f, sm = 0.0f;
    const int tid = fresh_tid();
    float* zre = reinterpret_cast<float*>(lds + L::R_OFF);
    float* zim = zre + G::ROWS * G::ZS;
    const bool aligned = ((reinterpret_cast<uintptr_t>(ref) | reinterpret_cast<uintptr_t>(img)) & 15) == 0;
    if constexpr (!FOLD && sizeof(TIn) == 4) if (ny == 64 && nx == 64 && aligned) {
        // full tiles: 16-byte global loads (coalesced 1 KiB per wave-instruction) and
        // 16-byte LDS stores; the image row is read back to front for the flip
        const f32x4* r4 = reinterpret_cast<const f32x4*>(ref);
        const f32x4* m4 = reinterpret_cast<const f32x4*>(img);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * kThreads;            // 1024 float4 per image
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
            for (int e = 0; e < 4; ++e) {
                zre[y * G::ZS + G::perm(idx & 15, e)] = r[e];
                zim[y * G::ZS + G::perm(idx & 15, e)] = m[e];
                sr += r[e] * r[e];
                sm += m[e] * m[e];
            }
        }
        ssq[0] = sr;
        ssq[1] = sm;
        return;
    }
    // every other shape: 4-pixel chunks, one 16-byte load each (4-byte aligned is enough on
    // gfx950), all issued before the first is used (chunk_issue); zeros in the padding; the
    // image is read back to front (cc.py:114)
    constexpr int kIters = (G::ROWS * G::CHUNKS + kThreads - 1) / kThreads;
    ChunkLoad<TIn> ld[kIters];
#pragma unroll
    for (int i = 0; i < kIters; ++i) {
        const int idx = tid + i * kThreads;
        const int y = FOLD ? idx / G::CHUNKS : idx >> 4;
        const int x = (FOLD ? idx - y * G::CHUNKS : (idx & 15)) << 2;
        ld[i] = chunk_issue<TIn, NARROW, NX4>(ref, img, ny, nx, y, x);        // (rows beyond the staged region: y >= ny, zeros)
    }
#pragma unroll
    for (int i = 0; i < kIters; ++i) {
        const int idx = tid + i * kThreads;
        if (FOLD && idx >= G::ROWS * G::CHUNKS) break;
        const int y = FOLD ? idx / G::CHUNKS : idx >> 4;
        const int x = (FOLD ? idx - y * G::CHUNKS : (idx & 15)) << 2;
        float rr[4], mm[4];
        chunk_unpack(ld[i], ns, rr, mm);
        if constexpr (FOLD) {
            *reinterpret_cast<f32x4*>(zre + y * G::ZS + x) = f32x4{rr[0], rr[1], rr[2], rr[3]};
            *reinterpret_cast<f32x4*>(zim + y * G::ZS + x) = f32x4{mm[0], mm[1], mm[2], mm[3]};
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                zre[y * G::ZS + G::perm(x >> 2, e)] = rr[e];
                zim[y * G::ZS + G::perm(x >> 2, e)] = mm[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { sr += rr[e] * rr[e]; sm += mm[e] * mm[e]; }
    }
    ssq[0] = sr;
    ssq[1] = sm;
}
template <int C, bool FOLD = false, typename TIn = float>
SPX_DEVICE void stage_pair(unsigned char* lds, const TIn* __restrict__ ref,
                           const TIn* __restrict__ img, int ny, int nx,
                           const NormStatsT<TIn>& ns, float (&ssq)[2]) {
    if (nx < 4) stage_pair_rows<C, FOLD, TIn, true>(lds, ref, img, ny, nx, ns, ssq);     // (uniform per item)
    else if ((nx & 3) == 0) stage_pair_rows<C, FOLD, TIn, false, true>(lds, ref, img, ny, nx, ns, ssq);
    else stage_pair_rows<C, FOLD, TIn, false>(lds, ref, img, ny, nx, ns, ssq);
}

// Z = FFT(ref + i*bal*flip(img)) is squared, so the cross term 2 ref*img is rounded
// relative to ref*ref + img*img: keep the two images at comparable amplitude.  `bal`
// is an exact power of two within a factor 2 of sqrt(sum ref^2 / sum img^2) (1 when either
// is zero); results are multiplied by 1/bal, also exact.  Workgroup-wide; one barrier.
// bal = 2^floor((E0 - E1)/2) from the exponent fields alone (no division, no overflow
// of the ratio: counts against counts/s may differ by 1e30 and more), clamped to 2^+-100.
SPX_DEVICE float balance_from_ssq(float s0, float s1) {
    if (!(s0 > 0.0f) || !(s1 > 0.0f)) return 1.0f;
    const int e0 = (int)((__builtin_bit_cast(unsigned, s0) >> 23) & 0xffu);
    const int e1 = (int)((__builtin_bit_cast(unsigned, s1) >> 23) & 0xffu);
    int e = (e0 - e1) >> 1;
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    return __builtin_bit_cast(float, (unsigned)(127 + e) << 23);
}
SPX_DEVICE float balance_factor(unsigned char* lds_scr, float (&ssq)[2]) {
    block_sum2f(lds_scr, ssq[0], ssq[1]);
    return balance_from_ssq(ssq[0], ssq[1]);
}

// Physical column of plane element (row, col): an XOR swizzle on column bits 3-4 by the
// row's low two bits.  It makes the plane writes of cc_planes (4 rows x 8 columns per
// 32-lane group) and the 2-row MFMA operand reads bank-conflict free, keeps groups of 8
// columns contiguous (16-byte reads stay aligned), and costs two integer ops.
SPX_DEVICE int plane_col(int row, int col) { return col ^ (((row & 1) << 4) | ((row & 2) << 2)); }

// ---------------------------------------------------------------------------
// Structure-of-arrays operands (rt::bcast_mul / rt::bcast_fma_rot): element K (0..3) of a 16-byte
// read `re4` of real parts and `im4` of imaginary parts, times w -- two packed instructions, and the
// result is the (re, im) pair the butterflies want.  Wherever a multiply follows such a read anyway
// (class twiddle after the tile load, stage twiddle after the transposition) this replaces the
// moves that used to build the pair first (~190 of a wave's ~1350 non-arithmetic vector
// instructions per pair).
// ---------------------------------------------------------------------------
// Reference mode's eight-transform kernel with transpose_tile_cplx: 0 spilled registers instead of 4 (235-237
// VGPRs) and 0.4-0.9 % SLOWER on the fold path that uses it (profiles/r03/disp5_cplxt_ab.txt: the half-wave
// reads cost more than four spilled registers do).  Shipped: 0 = the faster one, with its 4 spills.
#ifndef SPX_DISP5_CPLXT
#define SPX_DISP5_CPLXT 0
#endif
// MEASURED AND NOT ADOPTED (SPX_FUSED_TW = 0 is the shipped path; -DSPX_FUSED_TW=1 builds the variant;
// profiles/r03/variants_b1t0_fused.txt, sq_summary_fused.json): 8.5 % fewer vector instructions per pair
// (16.3k -> 14.9k), the same 3.92 ms per 1e5 pairs -- the time went into waits instead (SQ_WAIT_INST_LDS
// +83 %, ten spilled registers around the transposition).  The 64-tile kernel is not bound by the number
// of vector instructions it issues.
#ifndef SPX_FUSED_TW
#define SPX_FUSED_TW 0
#endif
template <int K> SPX_DEVICE f32x2 quad_pair(f32x4 q) {
    if constexpr (K < 2) return f32x2{q[0], q[1]}; else return f32x2{q[2], q[3]};
}
// (re4[K] + i im4[K]) w        (CONJ: conj(w))
template <int K, bool CONJ> SPX_DEVICE cf soa_cmul_k(f32x4 re4, f32x4 im4, cf w) {
    return rt::bcast_fma_rot<(K & 1), CONJ>(quad_pair<K>(im4), w, rt::bcast_mul<(K & 1), CONJ>(quad_pair<K>(re4), w));
}
template <bool CONJ> SPX_DEVICE cf soa_cmul(int k, f32x4 re4, f32x4 im4, cf w) {
    switch (k & 3) {                      // k is a constant after unrolling
        case 0: return soa_cmul_k<0, CONJ>(re4, im4, w);
        case 1: return soa_cmul_k<1, CONJ>(re4, im4, w);
        case 2: return soa_cmul_k<2, CONJ>(re4, im4, w);
        default: return soa_cmul_k<3, CONJ>(re4, im4, w);
    }
}
// re4[K] w + i im4[K] bw: the staged operand (re, bal im) times w with bw = bal w
template <int K> SPX_DEVICE cf soa_cmul2_k(f32x4 re4, f32x4 im4, cf w, cf bw) {
    return rt::bcast_fma_rot<(K & 1), false>(quad_pair<K>(im4), bw, rt::bcast_mul<(K & 1), false>(quad_pair<K>(re4), w));
}
SPX_DEVICE cf soa_cmul2(int k, f32x4 re4, f32x4 im4, cf w, cf bw) {
    switch (k & 3) {
        case 0: return soa_cmul2_k<0>(re4, im4, w, bw);
        case 1: return soa_cmul2_k<1>(re4, im4, w, bw);
        case 2: return soa_cmul2_k<2>(re4, im4, w, bw);
        default: return soa_cmul2_k<3>(re4, im4, w, bw);
    }
}

// The lane <-> register transposition (transpose_tile) with the reads left in structure-of-arrays
// form and the twiddles of the register's second digit applied on the way out:
//   v'[a][b] = T(v)[a][b] * w[b]      (CONJ: conj(w[b]);  W0_ONE: w[0] = 1, column 0 is only re-paired)
template <int XS, bool CONJ, bool W0_ONE>
SPX_DEVICE void transpose_tile_tw(cf (&v)[8][8], float* xch, int lane, const cf (&w)[8]) {
    f32x4 re4[16], im4[16];
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int r = 0; r < 64; ++r)
            xch[r * XS + lane] = part ? v[r >> 3][r & 7].y : v[r >> 3][r & 7].x;
        rt::wave_sync();
        const f32x4* row = reinterpret_cast<const f32x4*>(xch + lane * XS);
#pragma unroll
        for (int j = 0; j < 16; ++j) { if (part) im4[j] = row[j]; else re4[j] = row[j]; }
        rt::wave_sync();
    }
#pragma unroll
    for (int r = 0; r < 64; ++r) {
        const int b = r & 7;
        if (W0_ONE && b == 0) v[r >> 3][b] = cf{re4[r >> 2][r & 3], im4[r >> 2][r & 3]};
        else v[r >> 3][b] = soa_cmul<CONJ>(r, re4[r >> 2], im4[r >> 2], w[b]);
    }
}

// Tile load of the 64 tile with the class pre-twiddle w_P^{8 (CY y1 + CX x1)} (the free radix-2 stage
// of the zero pad; wave-uniform values) folded into the step that pairs the staged real and imaginary
// parts: v[y1][x1] = (re + i bal im) W_k, k = CY y1 + CX x1.  Class (1,1) used to pay 112 complex
// multiplies on top of the pairing moves; now every class costs the same two instructions per element.
template <typename G, int CY, int CX>
SPX_DEVICE void load_tile_class(cf (&v)[8][8], const float* zre, const float* zim, const cf* tw, float bal,
                                int l1, int l0) {
    constexpr int NK = 7 * (CY + CX) + 1;
    cf W[NK], BW[NK];
#pragma unroll
    for (int k = 1; k < NK; ++k) { W[k] = tw[8 * k]; BW[k] = cf{bal * W[k].x, bal * W[k].y}; }
#pragma unroll
    for (int y1 = 0; y1 < 8; ++y1) {
        const f32x4* pr = reinterpret_cast<const f32x4*>(zre + (l1 + 8 * y1) * G::ZS + l0 * G::PS12);
        const f32x4* pi = reinterpret_cast<const f32x4*>(zim + (l1 + 8 * y1) * G::ZS + l0 * G::PS12);
        const f32x4 r0 = pr[0], r1 = pr[1], i0 = pi[0], i1 = pi[1];
#pragma unroll
        for (int x1 = 0; x1 < 8; ++x1) {
            const int k = CY * y1 + CX * x1;
            const f32x4 rq = x1 < 4 ? r0 : r1, iq = x1 < 4 ? i0 : i1;
            if (k == 0) v[y1][x1] = cf{rq[x1 & 3], bal * iq[x1 & 3]};
            else v[y1][x1] = soa_cmul2(x1, rq, iq, W[k], BW[k]);
        }
    }
}

// The class plane: Im(v conj(W_k)) with the class post-twiddle W_k = w_P^{8 (CY y1 + CX x1)} (wave-uniform
// values: one multiply and one fused multiply-add per element; nothing at all for class (0,0)).
template <typename L, int CY, int CX>
SPX_DEVICE void store_plane_class(const cf (&v)[8][8], float* plane, const cf* tw, int l1, int l0) {
    constexpr int NK = 7 * (CY + CX) + 1;
    cf W[NK];
#pragma unroll
    for (int k = 1; k < NK; ++k) W[k] = tw[8 * k];
#pragma unroll
    for (int x1 = 0; x1 < 8; ++x1)
#pragma unroll
        for (int y1 = 0; y1 < 8; ++y1) {
            const cf a = v[y1][x1];
            const int k = CY * y1 + CX * x1;
            const int row = l1 + 8 * y1;
            plane[row * L::PS + plane_col(row, l0 + 8 * x1)] = k == 0 ? a.y : a.y * W[k].x - a.x * W[k].y;
        }
}

// Diagnostic early exit for phase timing (tools/phase_timing.py, `make diag`): DBG
// is a template parameter, 0 in the product library, so production code carries none
// of it.  A stopped variant folds its registers into one float per lane and stores
// it, which keeps the compiler from deleting the work done so far.
SPX_DEVICE float fold_tile(cf (&v)[8][8]) {
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 64; ++r) s += v[r >> 3][r & 7].x + v[r >> 3][r & 7].y;
    return s;
}
#define SPX_DBG_STOP(k)                                                        \
    do {                                                                       \
        if constexpr (DBG == (k)) {                                            \
            reinterpret_cast<float*>(lds + L::SCR_OFF + 768)[tid & 63] = fold_tile(v); \
            return true;                                                       \
        }                                                                      \
    } while (0)

// ---------------------------------------------------------------------------
// cc_planes: staged input planes -> the NCLS real class planes d_c in LDS.
// Caller must have issued a block_sync after staging; ends with a block_sync.
// ---------------------------------------------------------------------------
// CPLXT: transpose_tile_cplx (whole complex elements, no re-pairing moves, ~19 registers fewer) instead of
// transpose_tile: for callers that sit at the 256-register budget (reference mode: its spills go away).
template <int C, int DBG = 0, bool FOLD = false, bool CPLXT = false>
SPX_DEVICE bool cc_planes(unsigned char* lds, float bal, PhaseClock<DBG>& clk, int rot = 0) {
    typedef Lds<C> L;
    typedef StageGeom<C, FOLD> G;
    static_assert(C == 2, "class decomposition implemented for P = 128");
    const int tid = fresh_tid();
    const int lane = tid & 63;
    // This wave's parity class.  Classes differ in twiddle work ((0,0) has none, (1,1) the
    // most), so the pair kernel rotates the class <-> wave assignment from pair to pair
    // (`rot`); exchange buffer, class plane and fine window are indexed by CLASS, so nothing
    // downstream depends on which wave produced them.
    const int wave = ((tid >> 6) + rot) & (C * C - 1);
    const int cy = wave / C, cx = wave % C;
    const int l1 = lane >> 3, l0 = lane & 7;      // lane digits (y-ish, x-ish)
    const cf* tw = reinterpret_cast<const cf*>(lds + L::TW_OFF);
    const float* zre = reinterpret_cast<const float*>(lds + L::R_OFF);
    const float* zim = zre + G::ROWS * G::ZS;
    float* xch = reinterpret_cast<float*>(lds + L::R_OFF + wave * L::XCH_WAVE_BYTES);

    cf v[8][8];

    // ---- forward round A: lane = (y0, x0), registers = (y1, x1); y = y0 + 8 y1
    // FOLD: the class's radix-2 fold of the samples beyond index 63 (they exist for
    // y, x < 24 only: the cutout ends before 88)
    const float fsx = cx ? -1.0f : 1.0f, fsy = cy ? -1.0f : 1.0f;
    if constexpr (!FOLD && SPX_FUSED_TW) {
        if (wave == 0) load_tile_class<G, 0, 0>(v, zre, zim, tw, bal, l1, l0);
        else if (wave == 1) load_tile_class<G, 0, 1>(v, zre, zim, tw, bal, l1, l0);
        else if (wave == 2) load_tile_class<G, 1, 0>(v, zre, zim, tw, bal, l1, l0);
        else load_tile_class<G, 1, 1>(v, zre, zim, tw, bal, l1, l0);
    }
    if constexpr (!FOLD && !SPX_FUSED_TW) {
        // permuted rows (StageGeom): the lane's eight samples of a row are 32 contiguous bytes per plane
#pragma unroll
        for (int y1 = 0; y1 < 8; ++y1) {
            const f32x4* pr = reinterpret_cast<const f32x4*>(zre + (l1 + 8 * y1) * G::ZS + l0 * G::PS12);
            const f32x4* pi = reinterpret_cast<const f32x4*>(zim + (l1 + 8 * y1) * G::ZS + l0 * G::PS12);
            const f32x4 r0 = pr[0], r1 = pr[1], i0 = pi[0], i1 = pi[1];
#pragma unroll
            for (int x1 = 0; x1 < 4; ++x1) {
                v[y1][x1] = cf{r0[x1], bal * i0[x1]};
                v[y1][x1 + 4] = cf{r1[x1], bal * i1[x1]};
            }
        }
    }
#pragma unroll
    for (int y1 = 0; y1 < (FOLD ? 8 : 0); ++y1)
#pragma unroll
        for (int x1 = 0; x1 < 8; ++x1) {
            const int a = (l1 + 8 * y1) * G::ZS + l0 + 8 * x1;
            float re = zre[a], im = zim[a];
            if constexpr (FOLD) {
                if (x1 < 3) {
                    re = __builtin_fmaf(fsx, zre[a + 64], re);
                    im = __builtin_fmaf(fsx, zim[a + 64], im);
                }
                if (y1 < 3) {
                    float re2 = zre[a + 64 * G::ZS], im2 = zim[a + 64 * G::ZS];
                    if (x1 < 3) {
                        re2 = __builtin_fmaf(fsx, zre[a + 64 * G::ZS + 64], re2);
                        im2 = __builtin_fmaf(fsx, zim[a + 64 * G::ZS + 64], im2);
                    }
                    re = __builtin_fmaf(fsy, re2, re);
                    im = __builtin_fmaf(fsy, im2, im);
                }
            }
            v[y1][x1] = cf{re, bal * im};
        }
    rt::block_sync_lds();                       // all waves have read the staged input
    clk.tick(1);
    rt::set_prio<0>();                          // the transforms are throughput work

    // class pre-twiddle w_P^{c (8 y1)} (the free radix-2 stage of the zero pad; the 64 tile's load has it)
    if (cy && (FOLD || !SPX_FUSED_TW)) {
#pragma unroll
        for (int y1 = 1; y1 < 8; ++y1) {
            const cf w = tw[8 * cy * y1];
            mul_row<false>(v, y1, w);
        }
    }
    if (cx && (FOLD || !SPX_FUSED_TW)) {
#pragma unroll
        for (int x1 = 1; x1 < 8; ++x1) {
            const cf w = tw[8 * cx * x1];
            mul_col<false>(v, x1, w);
        }
    }
    fft8_y<1>(v);                           // y1 -> kyb
    fft8_x<1>(v);                           // x1 -> kxb
    SPX_DBG_STOP(2);
    clk.tick(2);
    // twiddle w_P^{y0 (cy + C kyb)} w_P^{x0 (cx + C kxb)}
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        const cf wy = tw[l1 * (cy + C * kb)];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[kb][j] = cmul(v[kb][j], wy);
    }
    if constexpr (!SPX_FUSED_TW) {
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const cf wx = tw[l0 * (cx + C * kb)];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j][kb] = cmul(v[j][kb], wx);
        }
    }

    SPX_DBG_STOP(3);
    clk.tick(3);
    // ---- transposition: (lane (y0,x0), reg (kyb,kxb)) -> (lane (kyb,kxb), reg (y0,x0))
    if constexpr (SPX_FUSED_TW) {
        // ... and the x twiddle on the way out: there the lane holds kxb (= l0) and the register x0
        cf wxa[8];
#pragma unroll
        for (int x0 = 1; x0 < 8; ++x0) wxa[x0] = tw[x0 * (cx + C * l0)];
        wxa[0] = cf{1.0f, 0.0f};
        transpose_tile_tw<L::XS, false, true>(v, xch, lane, wxa);
    } else if constexpr (CPLXT) {
        transpose_tile_cplx<L::XS>(v, xch, lane);
    } else {
        transpose_tile_sel<L::XS>(v, xch, lane);
    }

    SPX_DBG_STOP(4);
    clk.tick(4);
    // ---- forward round B: lane = (kyb, kxb), registers (y0, x0) -> (kya, kxa)
    fft8_y<1>(v);
    fft8_x<1>(v);
    SPX_DBG_STOP(5);
    clk.tick(5);
    // now v[kya][kxa] = Z[ky][kx], ky = cy + C*l1 + 8C*kya, kx = cx + C*l0 + 8C*kxa

    // ---- spectral product: W = Z^2 (see the header: Im IFFT(Z^2) = 2 ref*flip(img))
#pragma unroll
    for (int ka = 0; ka < 8; ++ka)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            v[ka][kb] = cmul(v[ka][kb], v[ka][kb]);
        }

    SPX_DBG_STOP(6);
    clk.tick(6);
    // ---- inverse round A': registers (kya, kxa) -> (y0, x0), lane = (kyb, kxb)
    fft8_y<-1>(v);
    fft8_x<-1>(v);
#pragma unroll
    for (int y0 = 1; y0 < 8; ++y0) {          // y0 = 0: w^0
        const cf wy = tw[y0 * (cy + C * l1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[y0][j] = cmulc(v[y0][j], wy);
    }
    if constexpr (!SPX_FUSED_TW) {
#pragma unroll
        for (int x0 = 1; x0 < 8; ++x0) {          // x0 = 0: w^0
            const cf wx = tw[x0 * (cx + C * l0)];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j][x0] = cmulc(v[j][x0], wx);
        }
    }
    SPX_DBG_STOP(7);
    clk.tick(7);
    // ---- transposition back: -> lane (y0, x0), registers (kyb, kxb)
    if constexpr (SPX_FUSED_TW) {
        // ... with the x twiddle on the way out: the lane holds x0 (= l0) and the register kxb
        cf wxb[8];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) wxb[kb] = tw[l0 * (cx + C * kb)];
        transpose_tile_tw<L::XS, true, false>(v, xch, lane, wxb);
    } else if constexpr (CPLXT) {
        transpose_tile_cplx<L::XS>(v, xch, lane);
    } else {
        transpose_tile_sel<L::XS>(v, xch, lane);
    }
    SPX_DBG_STOP(8);
    clk.tick(8);
    // ---- inverse round B': registers (kyb, kxb) -> (y1, x1)
    fft8_y<-1>(v);
    fft8_x<-1>(v);
    SPX_DBG_STOP(9);
    clk.tick(9);
    rt::set_prio<1>();                          // plane write, arg-max, refine, fit: latency-bound
    // class post-twiddle conj(w_P^{8 (cy y1 + cx x1)}); the imaginary part is kept (the factor
    // 1/2 of conv = Im(IFFT(Z^2))/2 is folded into the readers' output scale).
    // The plane goes into this wave's own exchange buffer (its reads above are done).
    float* plane = reinterpret_cast<float*>(lds + L::R_OFF + wave * L::PLANE_STRIDE_BYTES);
    if constexpr (SPX_FUSED_TW) {
        if (wave == 0) store_plane_class<L, 0, 0>(v, plane, tw, l1, l0);
        else if (wave == 1) store_plane_class<L, 0, 1>(v, plane, tw, l1, l0);
        else if (wave == 2) store_plane_class<L, 1, 0>(v, plane, tw, l1, l0);
        else store_plane_class<L, 1, 1>(v, plane, tw, l1, l0);
    } else {
    if (cy) {
#pragma unroll
        for (int y1 = 1; y1 < 8; ++y1) {
            const cf wy = tw[8 * cy * y1];
            mul_row<true>(v, y1, wy);
        }
    }
#pragma unroll
    for (int x1 = 0; x1 < 8; ++x1) {
        const cf wx = cx ? tw[8 * cx * x1] : cf{1.0f, 0.0f};
#pragma unroll
        for (int y1 = 0; y1 < 8; ++y1) {
            const cf a = v[y1][x1];
            const int row = l1 + 8 * y1;
            plane[row * L::PS + plane_col(row, l0 + 8 * x1)] = a.y * wx.x - a.x * wx.y;      // Im(a conj w)
        }
    }
    }
    rt::block_sync_lds();
    clk.tick(10);
    return false;
}

// cross-correlation value at flipped 'same'-window index (qy, qx) of a (ny, nx)
// cutout.  scipy's 'same' crop starts at (n-1)//2 of the full convolution and
// cc.py:121-126 flips the window, so q <-> convolution index l = (n-1-q) + (n-1)//2
// (lag l - (n-1) = (n-1-q) - n//2).
SPX_DEVICE int conv_index(int n, int q) { return (n - 1 - q) + (n - 1) / 2; }

template <int C>
SPX_DEVICE float window_value(const unsigned char* lds, int ny, int nx, int qy, int qx,
                              float out_scale) {
    typedef Lds<C> L;
    const float* planes = reinterpret_cast<const float*>(lds + L::R_OFF);
    const int ly = conv_index(ny, qy), lx = conv_index(nx, qx);
    const int my = ly & 63, mx = lx & 63;
    const int sy = (ly >> 6) & 1, sx = (lx >> 6) & 1;
    static_assert(C == 2, "");
    float d[C * C];
#pragma unroll
    for (int c = 0; c < C * C; ++c)
        d[c] = planes[c * (L::PLANE_STRIDE_BYTES / 4) + my * L::PS + plane_col(my, mx)];
    // same association as coarse_argmax: d00 + fx d01 + fy (d10 + fx d11)
    const float fx = sx ? -1.0f : 1.0f, fy = sy ? -1.0f : 1.0f;
    const float acc = __builtin_fmaf(fy, __builtin_fmaf(fx, d[3], d[2]), __builtin_fmaf(fx, d[1], d[0]));
    return acc * out_scale;     // out_scale = 1 / (2 P^2 bal)
}

// ---------------------------------------------------------------------------
// statistics for cc.py:131-156 over `npool` images (1: pair mode, 4: 5-image)
// ---------------------------------------------------------------------------
template <typename TIn>
SPX_DEVICE NormStatsT<TIn> norm_stats(unsigned char* lds_scr, const TIn* __restrict__ ref,
                                      const TIn* __restrict__ ims, int npool, int64_t im_stride,
                                      int ny, int nx, int cc_type) {
    NormStatsT<TIn> ns;
    ns.active = 0;
    ns.im_mean = 0; ns.im_rstd = 1; ns.ref_mean = 0; ns.ref_rstd = 1;
    if (cc_type == CC_PLAIN) return ns;
    const int tid = fresh_tid();
    const int npx = ny * nx;
    // 16-byte loads when the layout allows it (images are im_stride apart, each npx pixels)
    bool vec = (npx & 3) == 0 && (im_stride & 3) == 0 &&
               ((reinterpret_cast<uintptr_t>(ref) | reinterpret_cast<uintptr_t>(ims)) & 15) == 0;
    const int nchunk = vec ? npx >> 2 : npx;        // loop units: 4 pixels or 1
    // ONE pass: counts, sums and sums of squares in float64 (pooled image pixels != 0; ref over
    // the union mask).  numpy's std is the population form (ddof = 0) about the mean; in
    // float64, sum(x^2)/n - mean^2 agrees with it to ~1e-16 * mean^2/var, far below float32.
    double a[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // n_im, sum_im, sum_im^2, n_union, sum_ref, sum_ref^2
#pragma unroll 2
    for (int i = tid; i < nchunk; i += kThreads) {
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
    block_sum<6>(lds_scr, a);
    const double n_im = a[0], n_un = a[3];
    const double im_mean = a[1] / n_im, ref_mean = a[4] / n_un;
    double b[2];
    b[0] = a[2] - a[1] * im_mean;       // sum (x - mean)^2 = sum x^2 - (sum x) mean
    b[1] = a[5] - a[4] * ref_mean;
    if (b[0] < 0.0) b[0] = 0.0;
    if (b[1] < 0.0) b[1] = 0.0;
    ns.active = 1;
    const bool zero = (cc_type == CC_ZNCC);
    ns.im_mean = zero ? (TIn)im_mean : (TIn)0;
    ns.im_rstd = (TIn)(1.0 / sqrt(b[0] / n_im));
    ns.ref_mean = zero ? (TIn)ref_mean : (TIn)0;
    ns.ref_rstd = (TIn)(1.0 / sqrt(b[1] / n_un));
    return ns;
}

// ---------------------------------------------------------------------------
// peak of a virtual NX x NY image given its arg-max and an accessor for values:
// centroid.py:158-236 for peak_fit_box=5 on images >= 5x5.  Thread 0 returns the
// result; `val(x, y)` is evaluated by threads 0..24.
// ---------------------------------------------------------------------------
template <typename ValFn>
SPX_DEVICE PeakResult peak_from_argmax(unsigned char* lds_scr, int imax, int jmax,
                                       int NX, int NY, ValFn val) {
    const int tid = rt::thread_id();
    PeakResult r;
    r.x = (double)imax; r.y = (double)jmax; r.status = ST_EDGE;
    if (imax == 0 || jmax == 0) return r;              // centroid.py:171-172
    int x1 = imax - 2, y1 = jmax - 2;                  // centroid.py:165-184
    if (x1 > NX - 5) x1 = NX - 5;
    if (y1 > NY - 5) y1 = NY - 5;
    if (x1 < 0) x1 = 0;
    if (y1 < 0) y1 = 0;
    double* fit = reinterpret_cast<double*>(lds_scr + SCR_FIT);
    if (tid < 25) fit[tid] = (double)val(x1 + tid % 5, y1 + tid / 5);
    rt::block_sync_lds();
    if (tid == 0) r = quad_fit_5x5(fit, x1, y1, imax, jmax, NX, NY);
    rt::block_sync_lds();
    return r;
}

// ---------------------------------------------------------------------------
// Fine (upsampled) window by MFMA (v_mfma_f32_16x16x4_f32, exact f32).
//   K_0(t) = 1/64 [1 + 2 sum_{j=1..31} cos(2 pi 2j t / P) + cos(2 pi 64 t / P)],
//   K_1(t) = 2/64 sum_{j odd, 1..63} cos(2 pi j t / P)                (P = 128)
// are the interpolation kernels of the two parity classes.  With the window centred
// on convolution index (lyc, lxc) and both plane axes addressed RELATIVE to it
// (m = lc + m'', m'' in [-32, 32)), the operands that hold K are the same for every
// pair, so they come from two lane-major constant tables (spx_tables.h):
//   kty[c][blk][lane][s]    = K_c(-(16 blk + lj - W/2)/U - (4 s + lk - 32))       stage 1, B
//   ktx[c][blk][lane][4t+r] = K_c(-(16 blk + lj - W/2)/U - (16 t + 4 lk + r - 32))  stage 2, A
// (lane = 16 lk + lj).  Computes, for fine offsets a', b' in [-W/2, W/2) around the
// flipped coarse index (qyc, qxc),  F[b][a] = conv interpolated at q = U*qc + offset,
// into fbuf[b*W + a] (first index = x offset).
// ---------------------------------------------------------------------------
// the lane-constant K operands of one wave (loaded early, under the coarse arg-max)
// Up to W = 32 they are held in registers; larger windows (upsample >= 28) fetch them from the
// L2-resident table where they are used, or the 32 WB registers would spill.
// The refine's arithmetic: float32 (v_mfma_f32_16x16x4_f32) or float64 (v_mfma_f64_16x16x4_f64; float64 tables,
// spx_tables.h make_ktab_f64).  Both take A[i = lane & 15][k = lane >> 4] and B[k = lane >> 4][j = lane & 15];
// they differ in where result row i lives: register r of lane group lk holds row 4 lk + r (float32) or
// lk + 4 r (float64) -- `drow`.  Stage 2 consumes stage 1's result registers directly as its B operand, so
// its table is laid out in the same order (make_ktab / make_ktab_f64).
// Why float64: like for like against the float64 definition the float64-refine family (period 192) stays at
// 1-5e-5 px up to upsample 40, the float32-refine tiles are at 5e-5 ... 2.2e-4 from upsample 10 on
// (profiles/r03/refine_precision.txt); the window values are rounded to float32 either way, it is the two
// 64-term float32 accumulation chains that cost the digits.
struct RefineF32 {
    typedef float S;
    typedef f32x4 V4;
    static constexpr int kPreloadMax = 2;        // window blocks whose tables are held in registers
    static SPX_DEVICE V4 zero() { return V4{0.f, 0.f, 0.f, 0.f}; }
    static SPX_DEVICE V4 mma(S a, S b, V4 c) { return rt::mfma_16x16x4(a, b, c); }
    static SPX_DEVICE int drow(int lk, int r) { return 4 * lk + r; }
};
struct RefineF64 {
    typedef double S;
    typedef rt::f64x4 V4;
    static constexpr int kPreloadMax = 1;        // 16 doubles per table slice: 64 registers for one block
    static SPX_DEVICE V4 zero() { return V4{0.0, 0.0, 0.0, 0.0}; }
    static SPX_DEVICE V4 mma(S a, S b, V4 c) { return rt::mfma_f64_16x16x4(a, b, c); }
    static SPX_DEVICE int drow(int lk, int r) { return lk + 4 * r; }
};
// pair_kernel takes the arithmetic as a template parameter and the library holds both forms (the caller
// chooses per call: SPX_REFINE_* in include/subpixal_hip.h).  SPX_REFINE64_F64 only says what SPX_REFINE_DEFAULT
// means for the 64 tile: float32 -- measured 14 % faster at 64 px / upsample 10 and 18x inside the 1e-3 px
// tolerance there (profiles/r03/refine64_throughput_ab.txt) -- unless a build is made with -DSPX_REFINE64_F64=1.
#ifndef SPX_REFINE64_F64
#define SPX_REFINE64_F64 0
#endif
constexpr bool kRefine64DefaultF64 = SPX_REFINE64_F64 != 0;

template <int WB, typename R = RefineF32> struct FineTables {
    typedef typename R::V4 V4;
    static constexpr bool kPreload = WB <= R::kPreloadMax;
    V4 ky[kPreload ? WB : 1][4], kx[kPreload ? WB : 1][4];
    const V4* kty;
    const V4* ktx;
    SPX_DEVICE V4 y(int ab, int s4) const { return kPreload ? ky[ab][s4] : kty[ab * 64 * 4 + s4]; }
    SPX_DEVICE V4 x(int bb, int t) const { return kPreload ? kx[bb][t] : ktx[bb * 64 * 4 + t]; }
};
template <int C, int WB, typename R>
SPX_DEVICE void load_fine_tables(FineTables<WB, R>& ft, const float* __restrict__ ktab, int rot = 0) {
    typedef typename R::V4 V4;
    const int tid = fresh_tid();
    const int wave = ((tid >> 6) + rot) & (C * C - 1), lane = tid & 63;     // class, see cc_planes
    const int cy = wave / C, cx = wave % C;
    ktab = rt::launder(ktab);
    // table slices of this lane: [2 (y|x)][2 (class)][WB][64 lanes][16] elements of R::S
    const V4* kty = reinterpret_cast<const V4*>(ktab) + ((size_t)(0 * 2 + cy) * WB * 64 + lane) * 4;
    const V4* ktx = reinterpret_cast<const V4*>(ktab) + ((size_t)(1 * 2 + cx) * WB * 64 + lane) * 4;
    ft.kty = kty;
    ft.ktx = ktx;
    if constexpr (FineTables<WB, R>::kPreload) {
#pragma unroll
        for (int b = 0; b < WB; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ft.ky[b][i] = kty[b * 64 * 4 + i];
                ft.kx[b][i] = ktx[b * 64 * 4 + i];
            }
    }
}

template <int C, int WB, typename R>
SPX_DEVICE void fine_window(unsigned char* lds, const FineTables<WB, R>& ft,
                            int ny, int nx, int qyc, int qxc, int rot = 0) {
    typedef Lds<C> L;
    typedef typename R::S S;
    typedef typename R::V4 V4;
    static_assert(C == 2, "");
    constexpr int W = 16 * WB;
    const int tid = fresh_tid();
    const int wave = ((tid >> 6) + rot) & (C * C - 1), lane = tid & 63;     // class, see cc_planes
    const int cy = wave / C, cx = wave % C;
    const int lk = lane >> 4, lj = lane & 15;
    const float* plane = reinterpret_cast<const float*>(lds + L::R_OFF + wave * L::PLANE_STRIDE_BYTES);
    float* fbuf = reinterpret_cast<float*>(lds + L::FB_OFF);
    const int lyc = conv_index(ny, qyc), lxc = conv_index(nx, qxc);

    // d_c[m] = (-1)^(c floor(m/64)) plane[m mod 64]: the sign of a wrapped row goes
    // into the K operand of that row, the sign of a wrapped column likewise.
    V4 acc[WB][4];
#pragma unroll
    for (int ab = 0; ab < WB; ++ab)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[ab][t] = R::zero();
    int col[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) col[t] = (lxc + 16 * t + lj - 32) & 63;

    // stage 1: G^T[mx''][a] = sum_my'' plane[(lyc+my'')&63][(lxc+mx'')&63] sgn_y K_cy[a][my'']
    // all 64 A fragments first (the registers of the FFT tile are free here): one LDS
    // latency instead of one per group of MFMAs
    float afrag[16][4];
#pragma unroll
    for (int step = 0; step < 16; ++step) {
        const int row = (lyc + 4 * step + lk - 32) & 63;
#pragma unroll
        for (int t = 0; t < 4; ++t) afrag[step][t] = plane[row * L::PS + plane_col(row, col[t])];
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        V4 kb[WB];
#pragma unroll
        for (int ab = 0; ab < WB; ++ab) kb[ab] = ft.y(ab, s4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int step = 4 * s4 + e;
            const int m = lyc + 4 * step + lk - 32;
            const S sgn = (cy && ((m >> 6) & 1)) ? (S)-1 : (S)1;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int ab = 0; ab < WB; ++ab)
                    acc[ab][t] = R::mma((S)afrag[step][t], sgn * kb[ab][e], acc[ab][t]);
            }
        }
    }
    // stage 2: F^T[b][a] = sum_mx'' sgn_x K_cx[b][mx''] G^T[mx''][a]; accumulator register
    // r of tile t is B-operand row k' = lane>>4 for mx'' = 16 t + R::drow(k', r) - 32.
    const S scale = (S)0.5 / (S)(L::P * L::P);
    // where block (bb, ab) of this class's window goes: one window per class, which the reader adds in fixed
    // order (fine_value), or -- large windows -- ONE buffer that the classes add to in fixed order
    auto put = [&](int bb, int ab, const V4& fv, int c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = bb * 16 + R::drow(lk, r), a = ab * 16 + lj;
            const float val = (float)(fv[r] * scale);
            if constexpr (L::fb_count(W) == C * C) fbuf[wave * W * W + b * W + a] = val;
            else { if (c == 0) fbuf[b * W + a] = val; else fbuf[b * W + a] += val; }
        }
    };
    // All WB x WB result tiles live at once, in both arithmetic types.  For float64 a block-at-a-time stage 2 (as
    // fine_window128: only WB result tiles live) was built to keep the accumulators inside the register file and
    // measured slower at EVERY window size although it spills less -- its table reads from L2 sit in a serial loop:
    // two blocks 7.6 vs 6.0 ms per 1e5 pairs (0 spills either way), three 12.0 vs 8.8 ms (0 vs 2..7 spilled
    // registers), four 26.2 vs 19.7 ms (~52 vs ~143) -- profiles/r03/f64_live3_ab.txt, f64_live4_ab.txt.
    V4 f[WB][WB];
#pragma unroll
    for (int bb = 0; bb < WB; ++bb)
#pragma unroll
        for (int ab = 0; ab < WB; ++ab) f[bb][ab] = R::zero();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        V4 ka[WB];
#pragma unroll
        for (int bb = 0; bb < WB; ++bb) ka[bb] = ft.x(bb, t);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = lxc + 16 * t + R::drow(lk, r) - 32;
            const S sgn = (cx && ((m >> 6) & 1)) ? (S)-1 : (S)1;
#pragma unroll
            for (int bb = 0; bb < WB; ++bb)
#pragma unroll
                for (int ab = 0; ab < WB; ++ab)
                    f[bb][ab] = R::mma(sgn * ka[bb][r], acc[ab][t][r], f[bb][ab]);
        }
    }
    if constexpr (L::fb_count(W) == C * C) {
#pragma unroll
        for (int bb = 0; bb < WB; ++bb)
#pragma unroll
            for (int ab = 0; ab < WB; ++ab) put(bb, ab, f[bb][ab], 0);
        rt::block_sync_lds();
    } else {
        for (int c = 0; c < C * C; ++c) {
            if (wave == c) {
#pragma unroll
                for (int bb = 0; bb < WB; ++bb)
#pragma unroll
                    for (int ab = 0; ab < WB; ++ab) put(bb, ab, f[bb][ab], c);
            }
            rt::block_sync_lds();
        }
    }
}

// value of the fine window at (x offset index b, y offset index a)
template <int C, int W>
SPX_DEVICE float fine_value(const unsigned char* lds, int b, int a) {
    typedef Lds<C> L;
    const float* fbuf = reinterpret_cast<const float*>(lds + L::FB_OFF);
    if constexpr (L::fb_count(W) == C * C) {
        float acc = fbuf[b * W + a];
#pragma unroll
        for (int c = 1; c < C * C; ++c) acc += fbuf[c * W * W + b * W + a];
        return acc;
    } else {
        return fbuf[b * W + a];
    }
}

// ---------------------------------------------------------------------------
// Coarse arg-max over the flipped 'same' window, walking the class planes in
// storage order with 16-byte LDS reads.  A plane element (my, mx) is convolution
// index l = m or m + 64, whichever lies in the window [lo, lo + n), lo = (n-1)/2;
// its flipped window index is q = (n-1) + lo - l (see conv_index).
// ---------------------------------------------------------------------------
// FOLD (cutouts of 65..85 px): the window [lo, lo + n) is longer than a plane, so a plane
// element stands for up to two convolution indices per axis, m (sign +) and m + 64 (sign of
// the odd classes flipped); all four combinations are formed from the same four loads.
template <int C>
SPX_DEVICE void coarse_argmax_fold(const unsigned char* lds, int ny, int nx, float& bv, int& bi) {
    typedef Lds<C> L;
    static_assert(C == 2, "");
    const int tid = fresh_tid();
    const int loy = (ny - 1) / 2, lox = (nx - 1) / 2;
    const int mx4 = (tid & 15) << 2;
    // flipped window index of candidate A (l = m) and B (l = m + 64); negative = outside
    int qxa[4], qxb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int mx = mx4 + e;
        qxa[e] = mx >= lox ? (nx - 1) + lox - mx : -1;
        qxb[e] = (nx - 1) + lox - mx - 64;
    }
    float m = -__builtin_inff();
    int best = kNoIndex;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int my = (tid >> 4) + 16 * i;
        const int qya = my >= loy ? (ny - 1) + loy - my : -1;
        const int qyb = (ny - 1) + loy - my - 64;
        f32x4 d[C * C];
#pragma unroll
        for (int c = 0; c < C * C; ++c)
            d[c] = *reinterpret_cast<const f32x4*>(lds + L::R_OFF + c * L::PLANE_STRIDE_BYTES +
                                                   (my * L::PS + plane_col(my, mx4)) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // same association as window_value: (d00 + fx d01) + fy (d10 + fx d11)
            const float up = d[0][e] + d[1][e], um = d[0][e] - d[1][e];
            const float tp = d[2][e] + d[3][e], tm = d[2][e] - d[3][e];
            const float vaa = up + tp, vab = um + tm, vba = up - tp, vbb = um - tm;   // [y][x]
            if (qya >= 0 && qxa[e] >= 0 && better(vaa, qya * nx + qxa[e], m, best)) { m = vaa; best = qya * nx + qxa[e]; }
            if (qya >= 0 && qxb[e] >= 0 && better(vab, qya * nx + qxb[e], m, best)) { m = vab; best = qya * nx + qxb[e]; }
            if (qyb >= 0 && qxa[e] >= 0 && better(vba, qyb * nx + qxa[e], m, best)) { m = vba; best = qyb * nx + qxa[e]; }
            if (qyb >= 0 && qxb[e] >= 0 && better(vbb, qyb * nx + qxb[e], m, best)) { m = vbb; best = qyb * nx + qxb[e]; }
        }
    }
    bv = m;
    bi = best;
}

template <int C, bool FOLD = false>
SPX_DEVICE void coarse_argmax(const unsigned char* lds, int ny, int nx, float& bv, int& bi) {
    typedef Lds<C> L;
    static_assert(C == 2, "");
    if constexpr (FOLD) { coarse_argmax_fold<C>(lds, ny, nx, bv, bi); return; }
    const int tid = fresh_tid();
    const int loy = (ny - 1) / 2, lox = (nx - 1) / 2;
    const float ninf = -__builtin_inff();
    // column quantities of this thread's four elements (the same in all four row chunks)
    const int mx4 = (tid & 15) << 2;
    int qx[4];
    float fx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int mx = mx4 + e;
        const bool wrap = mx < lox;                 // convolution index mx + 64
        qx[e] = (nx - 1) + lox - mx - (wrap ? 64 : 0);
        fx[e] = wrap ? -1.0f : 1.0f;                // sign of the odd-x classes there
    }
    // values: d00 + fx d01 + fy (d10 + fx d11), unscaled (the scale is a positive power
    // of two); elements outside the window become -inf
    float val[4][4];
    int rowbase[4];
    float m = ninf;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int my = (tid >> 4) + 16 * i;         // 1024 float4 chunks per plane
        const bool wrap = my < loy;
        const int qy = (ny - 1) + loy - my - (wrap ? 64 : 0);
        const float fy = wrap ? -1.0f : 1.0f;
        rowbase[i] = qy * nx;
        f32x4 d[C * C];
#pragma unroll
        for (int c = 0; c < C * C; ++c)
            d[c] = *reinterpret_cast<const f32x4*>(lds + L::R_OFF + c * L::PLANE_STRIDE_BYTES +
                                                   (my * L::PS + plane_col(my, mx4)) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = __builtin_fmaf(fx[e], d[1][e], d[0][e]);
            const float t = __builtin_fmaf(fx[e], d[3][e], d[2][e]);
            const float v = __builtin_fmaf(fy, t, u);
            val[i][e] = (qy >= 0 && qx[e] >= 0) ? v : ninf;
            m = __builtin_fmaxf(m, val[i][e]);
        }
    }
    // smallest window index among this thread's elements equal to its maximum
    int best = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = (val[i][e] == m) ? rowbase[i] + qx[e] : 0x7fffffff;
            best = idx < best ? idx : best;
        }
    bv = m;
    bi = best;
}

// 5x5 fit by ONE wave (`fit_wave`, default 0) alone, no workgroup barrier, no LDS: its lanes
// 0..24 fetch the box values, quad_fit_wave does the rest.  The other waves return at once;
// the result is valid on every lane of the fitting wave.
template <typename ValFn>
SPX_DEVICE PeakResult peak_fit_wave0(unsigned char* lds_scr, int imax, int jmax, int NX, int NY,
                                     ValFn val, int fit_wave = 0) {
    const int tid = rt::thread_id() - 64 * fit_wave;
    PeakResult r;
    r.x = (double)imax; r.y = (double)jmax; r.status = ST_EDGE;
    if (imax == 0 || jmax == 0) return r;              // centroid.py:171-172
    if (tid < 0 || tid >= 64) return r;
    int x1 = imax - 2, y1 = jmax - 2;                  // centroid.py:165-184
    if (x1 > NX - 5) x1 = NX - 5;
    if (y1 > NY - 5) y1 = NY - 5;
    if (x1 < 0) x1 = 0;
    if (y1 < 0) y1 = 0;
    const int k = tid < 25 ? tid : 24;
    const float v = val(x1 + k % 5, y1 + k / 5);
    return quad_fit_wave(v, tid, x1, y1, imax, jmax, NX, NY);
}

// ---------------------------------------------------------------------------
// Pair kernel: one workgroup per (ref, img) pair.
//   out[2*pair + {0,1}] = (dx, dy) float64, status[pair]
// ---------------------------------------------------------------------------
template <int C, int WB, int DBG = 0, bool FOLD = false, typename TIn = float, typename R = RefineF32>
SPX_DEVICE void pair_body(const TIn* __restrict__ ref, const TIn* __restrict__ img,
                          int ny, int nx, int U, int cc_type, const cf* __restrict__ tw_g,
                          const float* __restrict__ ktab, double* __restrict__ out,
                          int* __restrict__ status, unsigned char* lds, PhaseClock<DBG>& clk,
                          const TIn* __restrict__ next_ref, const TIn* __restrict__ next_img,
                          float& warm, int fit_wave, double inv_u) {
    typedef Lds<C> L;
    ny = rt::launder_uniform(ny);
    nx = rt::launder_uniform(nx);
    U = rt::launder_uniform(U);
    const int tid = fresh_tid();
    unsigned char* scr = lds + L::SCR_OFF;
    const NormStatsT<TIn> ns = norm_stats(scr, ref, img, 1, 0, ny, nx, cc_type);
    float ssq[2];
    rt::consume(warm);        // the warm-up load of this pair has landed (or was never issued)
    stage_pair<C, FOLD, TIn>(lds, ref, img, ny, nx, ns, ssq);
    const float bal = balance_factor(scr, ssq);       // includes the barrier after staging
    const float oscale = 0.5f / ((float)(L::P * L::P) * bal);
    clk.tick(0);
    if constexpr (DBG == 1) return;
    const int rot = (C * C - fit_wave) & (C * C - 1);     // the fitting wave takes the lightest class (0,0)
    if (cc_planes<C, DBG, FOLD>(lds, bal, clk, rot)) return;
    if constexpr (DBG == 10) return;
    // the refine stage's constant operands: issue the loads now, use them after the arg-max -- and BEFORE the
    // warm-up below: vmcnt counts in issue order, so a wait for the tables must not include the warm-up's
    // trip to HBM (+1.3 %, profiles/r03/variants_*.txt)
    FineTables<(WB > 0 ? WB : 1), R> ft;
    if constexpr (WB > 0) load_fine_tables<C, WB, R>(ft, ktab, rot);
    // pull the next pair into L2 while this one is in its tail (issuing it before the transforms instead
    // was measured 2 % slower)
    if constexpr (sizeof(TIn) == 4) if (next_ref) warm = warm_next_pair(next_ref, next_img);

    // coarse arg-max over the flipped 'same' window (centroid.py:114-116)
    float bv;
    int bi;
    coarse_argmax<C, FOLD>(lds, ny, nx, bv, bi);
    block_argmax(scr, bv, bi, 0);
    // A NaN or Inf pixel makes the whole correlation NaN and no element ever compares greater:
    // numpy.argmax then returns index 0 and find_peak its integer position (centroid.py:114,
    // 171-172); same result here, flagged ST_NONFINITE, and the refine stage is skipped
    // (workgroup-uniform: every thread holds the same bi).
    const bool nonfinite = bi == kNoIndex;
    if (nonfinite) bi = 0;
    int qyc = bi / nx, qxc = bi - (bi / nx) * nx;
    clk.tick(11);
    if constexpr (DBG == 11) { if (tid == 0) out[0] = (double)bi; return; }

    PeakResult pk;
    if (nonfinite) {
        pk.x = 0.0; pk.y = 0.0; pk.status = ST_NONFINITE;
    } else if constexpr (WB == 0) {
        pk = peak_fit_wave0(scr, qxc, qyc, nx, ny, [&](int x, int y) {
            return window_value<C>(lds, ny, nx, y, x, oscale);
        }, fit_wave);
    } else {
        constexpr int W = 16 * (WB > 0 ? WB : 1);
        const int NX = U * nx, NY = U * ny;
        int imax = 0, jmax = 0;
        bool inside = false;
        for (int iter = 0; iter < 4; ++iter) {
            fine_window<C, (WB > 0 ? WB : 1), R>(lds, ft, ny, nx, qyc, qxc, rot);
            clk.tick(12);
            if constexpr (DBG == 12) { if (tid == 0) out[0] = (double)fine_value<C, W>(lds, 0, 0); return; }
            // arg-max over the part of the window inside the virtual image: every wave scans
            // the whole window itself, so all waves hold the result without another barrier
            const int fx0 = U * qxc - W / 2, fy0 = U * qyc - W / 2;
            float fv = -__builtin_inff();
            int fi = 0x7fffffff;
#pragma unroll 4
            for (int i = (tid & 63); i < W * W; i += 64) {
                const int b = i / W, a = i % W;           // storage order (x offset major):
                const int idx = a * W + b;                // conflict-free reads; idx = (y, x) row-major
                const int gy = fy0 + a, gx = fx0 + b;
                const float val = fine_value<C, W>(lds, b, a);
                const bool in = gy >= 0 && gy < NY && gx >= 0 && gx < NX;
                if (in && better(val, idx, fv, fi)) { fv = val; fi = idx; }
            }
            wave_argmax(fv, fi);
            clk.tick(13);
            if constexpr (DBG == 13) { if (tid == 0) out[0] = (double)fi; return; }
            if (fi == kNoIndex) { imax = jmax = -1; break; }     // non-finite window (overflow)
            const int a = fi / W, b = fi % W;
            jmax = fy0 + a;
            imax = fx0 + b;
            // the 5x5 box (clamped into the image) must lie inside the window
            int x1 = imax - 2, y1 = jmax - 2;
            if (x1 > NX - 5) x1 = NX - 5;
            if (y1 > NY - 5) y1 = NY - 5;
            if (x1 < 0) x1 = 0;
            if (y1 < 0) y1 = 0;
            const bool okx = (x1 >= fx0 && x1 + 4 < fx0 + W) || imax == 0;
            const bool oky = (y1 >= fy0 && y1 + 4 < fy0 + W) || jmax == 0;
            if (okx && oky) { inside = true; break; }
            // move the window centre one coarse pixel towards the peak and redo.  The centre may
            // sit ONE PAST the last coarse sample (q = n): the fine image extends (U-1)/U of a pixel
            // beyond it, and a peak in that strip is bracketed from there.
            if (!okx) qxc += (b < W / 2) ? -1 : 1;
            if (!oky) qyc += (a < W / 2) ? -1 : 1;
            qxc = qxc < 0 ? 0 : (qxc > nx ? nx : qxc);
            qyc = qyc < 0 ? 0 : (qyc > ny ? ny : qyc);
            rt::block_sync_lds();      // every wave is done with this window before it is rebuilt
        }
        clk.tick(16);
        if (inside) {
            const int fx0 = U * qxc - W / 2, fy0 = U * qyc - W / 2;
            pk = peak_fit_wave0(scr, imax, jmax, NX, NY, [&](int x, int y) {
                return fine_value<C, W>(lds, x - fx0, y - fy0);
            }, fit_wave);
        } else if (imax < 0) {
            pk.x = 0.0; pk.y = 0.0; pk.status = ST_NONFINITE;
        } else {
            pk.x = (double)imax; pk.y = (double)jmax; pk.status = ST_WINDOW;
        }
    }
    clk.tick(17);
    if (tid == 64 * fit_wave) {
        // cc.py:89-93 with the interlace factor 2 replaced by U
        out[0] = pk.x * inv_u - (double)((nx - 1) / 2);
        out[1] = pk.y * inv_u - (double)((ny - 1) / 2);
        if (status) status[0] = pk.status;
    }
    clk.tick(14);
}

template <int C, int WB, int DBG = 0, bool FOLD = false, typename TIn = float, typename R = RefineF32>
SPX_TKERNEL(256) void pair_kernel(const TIn* __restrict__ ref, const TIn* __restrict__ img,
                                  int64_t nbatch, int ny, int nx, int U, int cc_type,
                                  const cf* __restrict__ tw_g, const float* __restrict__ ktab,
                                  double* __restrict__ out, int* __restrict__ status) {
    SPX_DYN_LDS(lds);
    load_twiddles<C>(lds, tw_g);
    PhaseClock<DBG> clk;
    clk.start();
    const int64_t stride = (int64_t)ny * nx;
    const bool full = !FOLD && sizeof(TIn) == 4 && ny == 64 && nx == 64 &&
        ((reinterpret_cast<uintptr_t>(ref) | reinterpret_cast<uintptr_t>(img)) & 15) == 0;
    const int64_t step = rt::grid_size();
    float warm = 0.0f;
    // the wave that does the (serial) 5x5 fit rotates from pair to pair, and starts
    // differently in neighbouring workgroups, so that no SIMD carries it every time
    int fit_wave = (int)(rt::block_id() & 3);
    // 1/U once, kept in scalar registers (exact for the reference's U = 2 and every power
    // of two; otherwise within one ulp of the division)
    const double inv_u = rt::read_lane(1.0 / (double)U, 0);
    for (int64_t p = first_item(rt::block_id(), step); p < nbatch; p += step) {
        const bool more = full && (p + step < nbatch);
        pair_body<C, WB, DBG, FOLD, TIn, R>(ref + p * stride, img + p * stride, ny, nx, U, cc_type, tw_g, ktab,
                              out + 2 * p, status ? status + p : nullptr, lds, clk,
                              more ? ref + (p + step) * stride : nullptr,
                              more ? img + (p + step) * stride : nullptr, warm, fit_wave, inv_u);
        fit_wave = (fit_wave + 1) & 3;
        // upsample > 1: the last workgroup-wide step of a pair is the class sum of the fine
        // window; after it every wave only reads that window (own arg-max, wave 0's fit),
        // which the next pair's staging does not touch, so the waves run on into the next
        // pair and meet again at its first barrier.  upsample = 1 fits on the planes, which
        // the staging overwrites.
        if constexpr (WB == 0) rt::block_sync_lds();
        clk.tick(15);
    }
    // diagnostic build: the per-phase cycle totals go to the tail of the status array
    if constexpr (DBG == 100)
        clk.flush(reinterpret_cast<unsigned long long*>(status + nbatch));
}

// ---------------------------------------------------------------------------
// 5-image reference mode (cc.find_displacement, cc.py:21-95): ref against the 4
// half-pixel dithers, interlaced 2x image, arg-max, 5x5 fit on the 2x grid.
// icc: float [2ny][2nx] per item in global memory (user output or workspace).
// ---------------------------------------------------------------------------
// One dither's flipped 'same' window -> its interlaced positions icc[2 qy + oy][2 qx + ox]
// (cc.py:121-126), walking the class planes in storage order exactly like coarse_argmax
// (16-byte LDS reads, 3 FMAs per element); (bv, bi) accumulate the arg-max over the
// interlaced image (index = row-major position in it).
// FOLD variant (65..85 px), see coarse_argmax_fold
template <int C>
SPX_DEVICE void interlace_window_fold(const unsigned char* lds, int ny, int nx, float out_scale,
                                      float* __restrict__ icc, int ox, int oy, float& bv, int& bi) {
    typedef Lds<C> L;
    static_assert(C == 2, "");
    const int tid = fresh_tid();
    const int loy = (ny - 1) / 2, lox = (nx - 1) / 2;
    const int NX = 2 * nx;
    const int mx4 = (tid & 15) << 2;
    int gxa[4], gxb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int mx = mx4 + e;
        const int qb = (nx - 1) + lox - mx - 64;
        gxa[e] = mx >= lox ? 2 * ((nx - 1) + lox - mx) + ox : -1;
        gxb[e] = qb >= 0 ? 2 * qb + ox : -1;
    }
    float m = bv;
    int best = bi;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int my = (tid >> 4) + 16 * i;
        const int qb = (ny - 1) + loy - my - 64;
        const int rowa = my >= loy ? (2 * ((ny - 1) + loy - my) + oy) * NX : -1;
        const int rowb = qb >= 0 ? (2 * qb + oy) * NX : -1;
        f32x4 d[C * C];
#pragma unroll
        for (int c = 0; c < C * C; ++c)
            d[c] = *reinterpret_cast<const f32x4*>(lds + L::R_OFF + c * L::PLANE_STRIDE_BYTES +
                                                   (my * L::PS + plane_col(my, mx4)) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float up = d[0][e] + d[1][e], um = d[0][e] - d[1][e];
            const float tp = d[2][e] + d[3][e], tm = d[2][e] - d[3][e];
            const float vaa = (up + tp) * out_scale, vab = (um + tm) * out_scale;
            const float vba = (up - tp) * out_scale, vbb = (um - tm) * out_scale;
            auto put = [&](int row, int gx, float v) {
                if (row >= 0 && gx >= 0) {
                    icc[row + gx] = v;
                    v = nan_as_inf(v);
                    if (better(v, row + gx, m, best)) { m = v; best = row + gx; }
                }
            };
            put(rowa, gxa[e], vaa);
            put(rowa, gxb[e], vab);
            put(rowb, gxa[e], vba);
            put(rowb, gxb[e], vbb);
        }
    }
    bv = m;
    bi = best;
}

template <int C, bool FOLD = false>
SPX_DEVICE void interlace_window(const unsigned char* lds, int ny, int nx, float out_scale,
                                 float* __restrict__ icc, int ox, int oy, float& bv, int& bi) {
    typedef Lds<C> L;
    static_assert(C == 2, "");
    if constexpr (FOLD) { interlace_window_fold<C>(lds, ny, nx, out_scale, icc, ox, oy, bv, bi); return; }
    const int tid = fresh_tid();
    const int loy = (ny - 1) / 2, lox = (nx - 1) / 2;
    const int NX = 2 * nx;
    const float ninf = -__builtin_inff();
    const int mx4 = (tid & 15) << 2;
    int gx[4];
    float fx[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int mx = mx4 + e;
        const bool wrap = mx < lox;
        const int qx = (nx - 1) + lox - mx - (wrap ? 64 : 0);
        gx[e] = qx >= 0 ? 2 * qx + ox : -1;
        fx[e] = wrap ? -1.0f : 1.0f;
    }
    float val[4][4];
    int rowbase[4];
    float m = ninf;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int my = (tid >> 4) + 16 * i;
        const bool wrap = my < loy;
        const int qy = (ny - 1) + loy - my - (wrap ? 64 : 0);
        const float fy = wrap ? -1.0f : 1.0f;
        rowbase[i] = qy >= 0 ? (2 * qy + oy) * NX : -1;
        f32x4 d[C * C];
#pragma unroll
        for (int c = 0; c < C * C; ++c)
            d[c] = *reinterpret_cast<const f32x4*>(lds + L::R_OFF + c * L::PLANE_STRIDE_BYTES +
                                                   (my * L::PS + plane_col(my, mx4)) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u = __builtin_fmaf(fx[e], d[1][e], d[0][e]);
            const float t = __builtin_fmaf(fx[e], d[3][e], d[2][e]);
            const float v = __builtin_fmaf(fy, t, u) * out_scale;
            const bool ok = rowbase[i] >= 0 && gx[e] >= 0;
            if (ok) icc[rowbase[i] + gx[e]] = v;
            val[i][e] = ok ? nan_as_inf(v) : ninf;
            m = __builtin_fmaxf(m, val[i][e]);
        }
    }
    int best = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = (val[i][e] == m) ? rowbase[i] + gx[e] : 0x7fffffff;
            best = idx < best ? idx : best;
        }
    if (better(m, best, bv, bi)) { bv = m; bi = best; }
}

template <int C, bool FOLD = false, typename TIn = float>
SPX_DEVICE void disp5_body(const TIn* __restrict__ ref, const TIn* __restrict__ im4,
                           int ny, int nx, int cc_type, const cf* __restrict__ tw_g,
                           float* __restrict__ icc, double* __restrict__ out,
                           int* __restrict__ status, unsigned char* lds) {
    typedef Lds<C> L;
    unsigned char* scr = lds + L::SCR_OFF;
    const int64_t stride = (int64_t)ny * nx;
    NormStatsT<TIn> ns = norm_stats(scr, ref, im4, 4, stride, ny, nx, cc_type);
    // workgroup-uniform values -> scalar registers: they live across the four dithers' transforms, where
    // every vector register is taken (6-11 spilled VGPRs before, tools/kernel_regs.py)
    ns.im_mean = rt::read_lane(ns.im_mean, 0);
    ns.im_rstd = rt::read_lane(ns.im_rstd, 0);
    ns.ref_mean = rt::read_lane(ns.ref_mean, 0);
    ns.ref_rstd = rt::read_lane(ns.ref_rstd, 0);
    ns.active = rt::read_lane(ns.active, 0);

    float bv = -__builtin_inff();
    int bi = 0x7fffffff;
    const int NX = 2 * nx, NY = 2 * ny;
    for (int q = 0; q < 4; ++q) {            // order 00, 10, 01, 11 (cc.py:114-117)
        const int ox = q & 1, oy = q >> 1;   // icc[oy::2, ox::2] = cc[::-1, ::-1]
        float ssq[2];
        stage_pair<C, FOLD, TIn>(lds, ref, im4 + q * stride, ny, nx, ns, ssq);
        const float bal = balance_factor(scr, ssq);
        const float oscale = 0.5f / ((float)(L::P * L::P) * bal);
        PhaseClock<0> noclk;
        cc_planes<C, 0, FOLD, SPX_DISP5_CPLXT>(lds, bal, noclk);
        interlace_window<C, FOLD>(lds, ny, nx, oscale, icc, ox, oy, bv, bi);
        rt::block_sync_lds();                    // planes are overwritten by the next stage
    }
    rt::block_sync();        // icc (GLOBAL memory) written above is read below by other waves
    block_argmax(scr, bv, bi, 0);
    // a NaN (ranked +inf, see nan_as_inf) or an overflowed correlation: the integer position of
    // the first one, as numpy.argmax + the integer fallbacks of find_peak give, flagged
    const bool nonfinite = !(bv < __builtin_inff());
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

// Batches of cutouts of DIFFERENT shapes (the reference's cutouts are bounding boxes + padding,
// one shape per source: cutout.py:159-175, 1023-1031).  Item p's reference cutout starts at element
// off[p] of `ref` (items packed back to back), its four dithers at 4 off[p] of `im4`, each
// shp[2p] x shp[2p+1] = (ny, nx) pixels, its interlaced image at 4 off[p] of `icc`.  Null tables:
// a uniform batch of (ny, nx) cutouts.  An item whose shape the kernel family launched for it cannot
// take (side below 3 or above `max_side`) is not computed: (NaN, NaN), status ST_SHAPE.
// Catalog mode (`skip_below` > 0, round 3): several kernel families are launched over the SAME table, each
// taking the items whose larger side lies in (skip_below, max_side] and passing over the others without
// writing anything (the caller pre-fills the results with "not measured").
struct ItemTable {
    const int64_t* off;
    const int* shp;
    int max_side;
    int skip_below;
};
struct ItemView {
    int64_t off;
    int ny, nx;
    bool ok;
    bool skip;       // not this launch's item: leave its results alone
};
SPX_DEVICE ItemView item_view(const ItemTable& t, int64_t p, int ny, int nx) {
    ItemView v;
    v.skip = false;
    if (t.off) {
        v.off = t.off[p];
        v.ny = t.shp[2 * p];
        v.nx = t.shp[2 * p + 1];
        v.ok = v.ny >= 3 && v.nx >= 3 && v.ny <= t.max_side && v.nx <= t.max_side;
        if (t.skip_below > 0) {
            const int side = v.ny > v.nx ? v.ny : v.nx;
            v.skip = !v.ok || side <= t.skip_below;
            v.ok = v.ok && !v.skip;
        }
    } else {
        v.off = p * ((int64_t)ny * nx);
        v.ny = ny;
        v.nx = nx;
        v.ok = true;
    }
    return v;
}
SPX_DEVICE void item_refused(double* __restrict__ out, int* __restrict__ status, int64_t p, bool writer) {
    if (writer) {
        const double nan = __builtin_nan("");
        out[2 * p] = nan;
        out[2 * p + 1] = nan;
        if (status) status[p] = ST_SHAPE;
    }
}

template <int C, bool FOLD = false, typename TIn = float>
SPX_TKERNEL(256) void disp5_kernel(const TIn* __restrict__ ref, const TIn* __restrict__ im4,
                                   int64_t nbatch, int ny, int nx, int cc_type,
                                   const cf* __restrict__ tw_g, float* __restrict__ icc,
                                   double* __restrict__ out, int* __restrict__ status, ItemTable items) {
    SPX_DYN_LDS(lds);
    load_twiddles<C>(lds, tw_g);
    for (int64_t p = first_item(rt::block_id(), rt::grid_size()); p < nbatch; p += rt::grid_size()) {
        const ItemView it = item_view(items, p, ny, nx);
        if (!it.ok) { if (!it.skip) item_refused(out, status, p, rt::thread_id() == 0); continue; }
        disp5_body<C, FOLD, TIn>(ref + it.off, im4 + 4 * it.off, it.ny, it.nx, cc_type, tw_g,
                      icc + 4 * it.off, out + 2 * p, status ? status + p : nullptr, lds);
        rt::block_sync_lds();
    }
}

}  // namespace spx
