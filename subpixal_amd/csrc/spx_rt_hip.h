// Device-side runtime layer for the kernels in spx_kernels.h: gfx950 only.
// (tests/cpu_emu/spx_rt_emu.h provides the same names for the CPU logic-check
// build used by the unit tests; the product library is built from this file.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spx {
namespace rt {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

#define SPX_DEVICE __device__ __forceinline__
#define SPX_KERNEL(nthreads) extern "C" __global__ __launch_bounds__(nthreads)
#define SPX_TKERNEL(nthreads) __global__ __launch_bounds__(nthreads, 2)
// 512-thread workgroups, two per CU: four waves per SIMD, at most 128 VGPRs
#define SPX_TKERNEL8(nthreads) __global__ __launch_bounds__(nthreads, 4)
// one wave per SIMD: up to 512 registers per lane (spx_kernels5.h keeps two class tiles resident)
#define SPX_TKERNEL1(nthreads) __global__ __launch_bounds__(nthreads, 1)
// all LDS lives in ONE dynamic region (cdna guide G17: keep the base 16-B aligned)
#define SPX_STATIC_LDS(type, name, count) __shared__ type name[count]
#define SPX_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]

SPX_DEVICE int thread_id() { return (int)threadIdx.x; }
SPX_DEVICE int64_t block_id() { return (int64_t)blockIdx.x; }
SPX_DEVICE int64_t grid_size() { return (int64_t)gridDim.x; }

// workgroup barrier (LDS + global visibility at workgroup scope)
SPX_DEVICE void block_sync() { __syncthreads(); }

// Workgroup barrier that orders LDS traffic only: global loads/stores issued before it
// (prefetches, result stores) stay in flight across it (no vmcnt(0) drain, unlike
// __syncthreads()).  Use block_sync() where another wave reads GLOBAL data written
// before the barrier.
SPX_DEVICE void block_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Ordering point between LDS accesses of the lanes of ONE wave (per-wave LDS
// regions).  A wave's DS instructions execute in issue order, so only the
// compiler has to be stopped from reordering across this point.
SPX_DEVICE void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Makes a (wave-uniform) pointer opaque to the optimiser at this point, so loads
// through it are not hoisted out of the per-pair loop into long-lived registers.
// The pointer is a GLOBAL one and keeps that address space through the asm: a generic
// pointer would turn the loads into flat_load, which also ticks lgkmcnt and so makes every
// LDS wait behind them sit out an L2 round trip.
template <typename T> SPX_DEVICE T* launder(T* p) {
    typedef T __attribute__((address_space(1)))* global_ptr;
    global_ptr g = (global_ptr)p;
    asm volatile("" : "+s"(g));
    return (T*)g;
}

// same for a pointer that lives in vector registers (derived from the work-item id)
template <typename T> SPX_DEVICE T* launder_lanes(T* p) {
    typedef T __attribute__((address_space(1)))* global_ptr;
    global_ptr g = (global_ptr)p;
    asm volatile("" : "+v"(g));
    return (T*)g;
}

SPX_DEVICE int launder_uniform(int v) {
    asm volatile("" : "+s"(v));
    return v;
}
// Same for a per-lane value: address arithmetic derived from it is redone where it
// is needed instead of being kept in registers across the whole per-pair loop.
SPX_DEVICE int launder_lane(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// shader-clock stamp for the diagnostic build (cdna guide section 7, in-kernel stamps)
SPX_DEVICE unsigned long long clock_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
SPX_DEVICE void atomic_add_u64(unsigned long long* p, unsigned long long v) { atomicAdd(p, v); }
SPX_DEVICE void atomic_min_i32(int* p, int v) { atomicMin(p, v); }
SPX_DEVICE void atomic_max_i32(int* p, int v) { atomicMax(p, v); }
SPX_DEVICE void atomic_add_i32(int* p, int v) { atomicAdd(p, v); }

// ---------------------------------------------------------------------------
// Packed complex arithmetic on (re, im) register pairs: one VOP3P instruction each,
// using op_sel (which half feeds which lane) and neg_lo/neg_hi instead of the
// v_mov/v_xor + scalar mul/fma sequences hipcc emits for the same expressions.
// ---------------------------------------------------------------------------
// a * w
SPX_DEVICE f32x2 cmul(f32x2 a, f32x2 w) {
    f32x2 r;      // one statement: hipcc pads separate asm statements with s_nop
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "=&v"(r) : "v"(a), "v"(w));
    return r;
}
// a * conj(w)
SPX_DEVICE f32x2 cmulc(f32x2 a, f32x2 w) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=&v"(r) : "v"(a), "v"(w));
    return r;
}
// a *= w and a *= conj(w) IN PLACE (result in a's own registers, one scratch pair): used
// where the multiply sits in a conditional block, so that no register permutation is
// needed where the block re-joins the path that skipped it.
SPX_DEVICE void cmul_ip(f32x2& a, f32x2 w) {
    f32x2 t;
    asm("v_pk_mul_f32 %1, %0, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %0, %2, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "+v"(a), "=&v"(t) : "v"(w));
}
SPX_DEVICE void cmulc_ip(f32x2& a, f32x2 w) {
    f32x2 t;
    asm("v_pk_mul_f32 %1, %0, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %0, %2, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "+v"(a), "=&v"(t) : "v"(w));
}
// eight in-place multiplies by one w as ONE asm block with two alternating scratch pairs.  Between
// two single-element blocks that were given the same scratch pair the compiler inserts an s_nop (its
// conservative rule for a register written by one inline-asm block and touched by the next: it cannot
// see that the writer is not a partial-register write); inside a block there is nothing to protect.
template <bool CONJ>
SPX_DEVICE void cmul8_ip(f32x2& a0, f32x2& a1, f32x2& a2, f32x2& a3, f32x2& a4, f32x2& a5, f32x2& a6, f32x2& a7,
                         f32x2 w) {
    f32x2 t0, t1;
#define SPX_CM_MUL(T, A) "v_pk_mul_f32 " T ", " A ", %10 op_sel_hi:[1,0]\n\t"
#define SPX_CM_FMA(T, A, NEG) "v_pk_fma_f32 " A ", " A ", %10, " T " op_sel:[1,1,0] op_sel_hi:[0,1,1] " NEG ":[1,0,0]\n\t"
#define SPX_CM_BODY(NEG)                                                                                     \
    SPX_CM_MUL("%8", "%0") SPX_CM_MUL("%9", "%1") SPX_CM_FMA("%8", "%0", NEG) SPX_CM_FMA("%9", "%1", NEG)    \
    SPX_CM_MUL("%8", "%2") SPX_CM_MUL("%9", "%3") SPX_CM_FMA("%8", "%2", NEG) SPX_CM_FMA("%9", "%3", NEG)    \
    SPX_CM_MUL("%8", "%4") SPX_CM_MUL("%9", "%5") SPX_CM_FMA("%8", "%4", NEG) SPX_CM_FMA("%9", "%5", NEG)    \
    SPX_CM_MUL("%8", "%6") SPX_CM_MUL("%9", "%7") SPX_CM_FMA("%8", "%6", NEG) SPX_CM_FMA("%9", "%7", NEG)
    if constexpr (CONJ)
        asm(SPX_CM_BODY("neg_hi")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t0), "=&v"(t1)
            : "v"(w));
    else
        asm(SPX_CM_BODY("neg_lo")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t0), "=&v"(t1)
            : "v"(w));
#undef SPX_CM_BODY
#undef SPX_CM_FMA
#undef SPX_CM_MUL
}
// Complex arithmetic on operands that arrive as STRUCTURE-OF-ARRAYS register pairs: a 16-byte LDS
// read of four real (or four imaginary) parts lands in four consecutive registers, i.e. two aligned
// pairs a = (p_j, p_j+1).  VOP3P's op_sel picks either half of a pair for BOTH result lanes, so a
// product of a complex number given as (re in one pair, im in another) with w costs the same two
// packed instructions as cmul on an (re, im) pair and no moves to build that pair first:
//   bcast_mul<S>(a, w)         = (a[S] w.x,  a[S] w.y)          CONJ: (a[S] w.x, -a[S] w.y)
//   bcast_fma_rot<S>(a, w, c)  = c + a[S] (-w.y, w.x)           CONJ: c + a[S] (w.y, w.x)
// so that (re + i im) w = bcast_fma_rot<S>(IM, w, bcast_mul<S>(RE, w)), and with CONJ the product
// with conj(w).
template <int S, bool CONJ = false> SPX_DEVICE f32x2 bcast_mul(f32x2 a, f32x2 w) {
    f32x2 r;
    if constexpr (S == 0 && !CONJ) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(w));
    if constexpr (S == 1 && !CONJ) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(w));
    if constexpr (S == 0 && CONJ) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(w));
    if constexpr (S == 1 && CONJ) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(w));
    return r;
}
template <int S, bool CONJ = false> SPX_DEVICE f32x2 bcast_fma_rot(f32x2 a, f32x2 w, f32x2 c) {
    f32x2 r;
    if constexpr (S == 0 && !CONJ) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
    if constexpr (S == 1 && !CONJ) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
    if constexpr (S == 0 && CONJ) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
    if constexpr (S == 1 && CONJ) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
    return r;
}
// d += t * w IN PLACE: two fused multiply-adds, no temporary
SPX_DEVICE void cmac_ip(f32x2& d, f32x2 t, f32x2 w) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "+v"(d) : "v"(t), "v"(w));
}
// s + (-i) d = (s.x + d.y, s.y - d.x)
SPX_DEVICE f32x2 add_mi(f32x2 s, f32x2 d) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(s), "v"(d));
    return r;
}
// s + (+i) d = (s.x - d.y, s.y + d.x)
SPX_DEVICE f32x2 add_pi(f32x2 s, f32x2 d) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(s), "v"(d));
    return r;
}
// -d + (-i) d = (d.y - d.x, -d.x - d.y)
SPX_DEVICE f32x2 neg_add_mi(f32x2 d) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,1]" : "=v"(r) : "v"(d));
    return r;
}
// -d + (+i) d = (-d.x - d.y, d.x - d.y)
SPX_DEVICE f32x2 neg_add_pi(f32x2 d) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,1] neg_hi:[1,0]" : "=v"(r) : "v"(d));
    return r;
}

// a * b + c per component
SPX_DEVICE f32x2 fma_pk(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// (a.y * b.x + c.x, a.x * b.y + c.y): with b = (s, -s) this is c + s (-i) a
SPX_DEVICE f32x2 fma_swap(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// v_permlane32_swap: lanes 32..63 of a trade places with lanes 0..31 of b, i.e. afterwards
// a = [a_lo, b_lo], b = [a_hi, b_hi] (no LDS).  The builtin lets hipcc place the wait states the
// instruction needs after a VALU write of either operand; its two results are taken as integers
// first (bit-casting the vector elements directly mis-assigned the second one with ROCm 7.2).
SPX_DEVICE void swap_halves(float& a, float& b) {
    const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
    const auto r = __builtin_amdgcn_permlane32_swap(ua, ub, false, false);
    const unsigned ra = r[0], rb = r[1];
    a = __builtin_bit_cast(float, ra);
    b = __builtin_bit_cast(float, rb);
}

// Wave priority for instruction arbitration inside a SIMD (s_setprio; 0 = default, 3 = highest).  The
// latency-bound phases of a pair (staging, arg-max, refine, fit: few instructions, each waiting on LDS or a
// barrier) run at SPX_PRIO_TAIL so that they are not queued behind the other workgroup's VALU-dense
// transform phases (-DSPX_PRIO_TAIL=0 builds the kernels without it: A/B knob, profiles/r03).
#ifndef SPX_PRIO_TAIL
#define SPX_PRIO_TAIL 2          // measured +2.2 % on the 64 tile (profiles/r03/variants_prio_tail_prio_xform.txt)
#endif
#ifndef SPX_PRIO_XFORM
#define SPX_PRIO_XFORM 0
#endif
template <int P> SPX_DEVICE void set_prio() { __builtin_amdgcn_s_setprio(P ? SPX_PRIO_TAIL : SPX_PRIO_XFORM); }

// forces `v` to be materialised here (and nothing else)
SPX_DEVICE void consume(float v) { asm volatile("" ::"v"(v)); }

// stops the instruction scheduler from moving anything across this point (used to
// keep independent butterflies from being interleaved into a register-pressure spike)
SPX_DEVICE void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// one 4-byte LDS read that the compiler may not merge with its neighbours into a wider one
SPX_DEVICE float lds_read_f32(const float* p) {
    return *(const volatile __attribute__((address_space(3))) float*)p;
}

// same, but also orders the wave's GLOBAL stores before its later loads (vmcnt drain)
SPX_DEVICE void wave_sync_mem() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

SPX_DEVICE float shfl_xor(float v, int m) { return __shfl_xor(v, m, 64); }
SPX_DEVICE int shfl_xor(int v, int m) { return __shfl_xor(v, m, 64); }
SPX_DEVICE double shfl_xor(double v, int m) { return __shfl_xor(v, m, 64); }

// DPP lane exchanges inside a row of 16 lanes (no LDS round trip, unlike ds_bpermute):
// 0 = lane^1, 1 = lane^2, 2 = mirror within 8 lanes, 3 = mirror within 16 lanes.
template <int STEP> struct DppCtrl {
    static constexpr int value = STEP == 0 ? 0xB1 : STEP == 1 ? 0x4E : STEP == 2 ? 0x141 : 0x140;
};
template <int STEP> SPX_DEVICE int row_xchg(int v) {
    return __builtin_amdgcn_update_dpp(0, v, DppCtrl<STEP>::value, 0xF, 0xF, false);
}
template <int STEP> SPX_DEVICE float row_xchg(float v) {
    return __builtin_bit_cast(float, row_xchg<STEP>(__builtin_bit_cast(int, v)));
}
SPX_DEVICE float read_lane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
SPX_DEVICE int read_lane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
SPX_DEVICE double read_lane(double v, int lane) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// v_mfma_f32_16x16x4_f32: lane l holds A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
// D[row=4*(l>>4)+r][col=l&15] in register r.  Exact f32 fma chain over k.
SPX_DEVICE f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// v_mfma_f64_16x16x4_f64: A/B as the f32 form (one f64 per lane), but D[row=(l>>4)+4*r][col=l&15]
// in register r (cdna_hip_programming.md: the f64 MFMA does not use the f32 C/D map).
SPX_DEVICE f64x4 mfma_f64_16x16x4(double a, double b, f64x4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

}  // namespace rt
}  // namespace spx
