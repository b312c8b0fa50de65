// TEST INFRASTRUCTURE ONLY -- CPU logic-check build of the HIP kernels.
//
// Provides the names of subpixal_amd/csrc/spx_rt_hip.h on a CPU: one std::thread
// per work-item, std::barrier for the workgroup barrier, per-wave barriers for
// wave-collectives (shuffles, MFMA), one static buffer for the LDS.  It lets the
// unit tests (and ASan/UBSan) run the very same kernel source that hipcc
// compiles for gfx950, to check index arithmetic before a kernel touches a GPU.
// It is never linked into the product library and is orders of magnitude slower.
#pragma once
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

namespace spx {
namespace rt {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

#define SPX_DEVICE inline __attribute__((always_inline))
#define SPX_KERNEL(nthreads) extern "C"
#define SPX_TKERNEL(nthreads)
#define SPX_TKERNEL8(nthreads)
#define SPX_TKERNEL1(nthreads)
#define SPX_STATIC_LDS(type, name, count) static type name[count]
#define SPX_DYN_LDS(name) unsigned char* name = ::spx::rt::emu().lds

struct WaveState {
    std::unique_ptr<std::barrier<>> bar;
    float fa[64];
    float fb[64];
    int ia[64];
    double da[64];
    double db[64];
};

struct EmuState {
    unsigned char* lds = nullptr;     // one heap block of exactly the launch's dynamic LDS size
    std::unique_ptr<std::barrier<>> block_bar;
    std::vector<WaveState> waves;
    int nthreads = 0;
    int64_t nblocks = 0;
};

inline EmuState& emu() {
    static EmuState s;
    return s;
}

struct ThreadCtx {
    int tid = 0;
    int64_t bid = 0;
};
inline ThreadCtx& ctx() {
    static thread_local ThreadCtx c;
    return c;
}

SPX_DEVICE int thread_id() { return ctx().tid; }
SPX_DEVICE int64_t block_id() { return ctx().bid; }
SPX_DEVICE int64_t grid_size() { return emu().nblocks; }
SPX_DEVICE void block_sync() { emu().block_bar->arrive_and_wait(); }
SPX_DEVICE void block_sync_lds() { emu().block_bar->arrive_and_wait(); }

inline WaveState& my_wave() { return emu().waves[ctx().tid >> 6]; }
SPX_DEVICE void wave_sync() { my_wave().bar->arrive_and_wait(); }
SPX_DEVICE void wave_sync_mem() { my_wave().bar->arrive_and_wait(); }

template <typename T> SPX_DEVICE T* launder(T* p) { return p; }
template <typename T> SPX_DEVICE T* launder_lanes(T* p) { return p; }
SPX_DEVICE int launder_lane(int v) { return v; }
SPX_DEVICE int launder_uniform(int v) { return v; }
SPX_DEVICE unsigned long long clock_stamp() { return 0; }
SPX_DEVICE void atomic_add_u64(unsigned long long* p, unsigned long long v) { (void)p; (void)v; }
// one workgroup at a time, threads are real: use atomics
SPX_DEVICE void atomic_min_i32(int* p, int v) { int o = __atomic_load_n(p, __ATOMIC_RELAXED); while (v < o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} }
SPX_DEVICE void atomic_max_i32(int* p, int v) { int o = __atomic_load_n(p, __ATOMIC_RELAXED); while (v > o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} }
SPX_DEVICE void atomic_add_i32(int* p, int v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
SPX_DEVICE f32x2 cmul(f32x2 a, f32x2 w) { return f32x2{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
SPX_DEVICE f32x2 cmulc(f32x2 a, f32x2 w) { return f32x2{a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y}; }
SPX_DEVICE void cmac_ip(f32x2& d, f32x2 t, f32x2 w) { const f32x2 p = cmul(t, w); d = f32x2{d.x + p.x, d.y + p.y}; }
SPX_DEVICE void cmul_ip(f32x2& a, f32x2 w) { a = cmul(a, w); }
SPX_DEVICE void cmulc_ip(f32x2& a, f32x2 w) { a = cmulc(a, w); }
template <bool CONJ>
SPX_DEVICE void cmul8_ip(f32x2& a0, f32x2& a1, f32x2& a2, f32x2& a3, f32x2& a4, f32x2& a5, f32x2& a6, f32x2& a7,
                         f32x2 w) {
    f32x2* a[8] = {&a0, &a1, &a2, &a3, &a4, &a5, &a6, &a7};
    for (int i = 0; i < 8; ++i) *a[i] = CONJ ? cmulc(*a[i], w) : cmul(*a[i], w);
}
template <int S, bool CONJ = false> SPX_DEVICE f32x2 bcast_mul(f32x2 a, f32x2 w) {
    const float x = S ? a.y : a.x;
    return f32x2{x * w.x, CONJ ? x * -w.y : x * w.y};
}
template <int S, bool CONJ = false> SPX_DEVICE f32x2 bcast_fma_rot(f32x2 a, f32x2 w, f32x2 c) {
    const float x = S ? a.y : a.x;
    return f32x2{std::fmaf(x, CONJ ? w.y : -w.y, c.x), std::fmaf(x, w.x, c.y)};
}
SPX_DEVICE f32x2 add_mi(f32x2 s, f32x2 d) { return f32x2{s.x + d.y, s.y - d.x}; }
SPX_DEVICE f32x2 add_pi(f32x2 s, f32x2 d) { return f32x2{s.x - d.y, s.y + d.x}; }
SPX_DEVICE f32x2 neg_add_mi(f32x2 d) { return f32x2{d.y - d.x, -d.x - d.y}; }
SPX_DEVICE f32x2 neg_add_pi(f32x2 d) { return f32x2{-d.x - d.y, d.x - d.y}; }
SPX_DEVICE f32x2 fma_pk(f32x2 a, f32x2 b, f32x2 c) { return f32x2{std::fmaf(a.x, b.x, c.x), std::fmaf(a.y, b.y, c.y)}; }
SPX_DEVICE f32x2 fma_swap(f32x2 a, f32x2 b, f32x2 c) { return f32x2{std::fmaf(a.y, b.x, c.x), std::fmaf(a.x, b.y, c.y)}; }
template <int P> SPX_DEVICE void set_prio() {}
SPX_DEVICE float lds_read_f32(const float* p) { return *p; }
SPX_DEVICE void consume(float v) { (void)v; }
SPX_DEVICE void sched_fence() {}

SPX_DEVICE float shfl_xor(float v, int m) {
    WaveState& w = my_wave();
    int lane = ctx().tid & 63;
    w.fa[lane] = v;
    w.bar->arrive_and_wait();
    float r = w.fa[lane ^ m];
    w.bar->arrive_and_wait();
    return r;
}
SPX_DEVICE int shfl_xor(int v, int m) {
    WaveState& w = my_wave();
    int lane = ctx().tid & 63;
    w.ia[lane] = v;
    w.bar->arrive_and_wait();
    int r = w.ia[lane ^ m];
    w.bar->arrive_and_wait();
    return r;
}

template <typename T> SPX_DEVICE T shfl_lane(T v, int src, T* slot) {
    WaveState& w = my_wave();
    int lane = ctx().tid & 63;
    slot[lane] = v;
    w.bar->arrive_and_wait();
    T r = slot[src];
    w.bar->arrive_and_wait();
    return r;
}
inline int row_src(int step, int lane) {
    return step == 0 ? lane ^ 1 : step == 1 ? lane ^ 2
         : step == 2 ? ((lane & ~7) | (7 - (lane & 7))) : ((lane & ~15) | (15 - (lane & 15)));
}
template <int STEP> SPX_DEVICE float row_xchg(float v) { return shfl_lane(v, row_src(STEP, ctx().tid & 63), my_wave().fa); }
template <int STEP> SPX_DEVICE int row_xchg(int v) { return shfl_lane(v, row_src(STEP, ctx().tid & 63), my_wave().ia); }
SPX_DEVICE float read_lane(float v, int lane) { return shfl_lane(v, lane, my_wave().fa); }
SPX_DEVICE int read_lane(int v, int lane) { return shfl_lane(v, lane, my_wave().ia); }
SPX_DEVICE double read_lane(double v, int lane) { return shfl_lane(v, lane, my_wave().da); }

// v_permlane32_swap: a = [a_lo, b_lo], b = [a_hi, b_hi]
SPX_DEVICE void swap_halves(float& a, float& b) {
    WaveState& w = my_wave();
    int lane = ctx().tid & 63;
    w.fa[lane] = a;
    w.fb[lane] = b;
    w.bar->arrive_and_wait();
    if (lane < 32) b = w.fa[lane + 32]; else a = w.fb[lane - 32];
    w.bar->arrive_and_wait();
}

SPX_DEVICE double shfl_xor(double v, int m) {
    WaveState& w = my_wave();
    int lane = ctx().tid & 63;
    w.da[lane] = v;
    w.bar->arrive_and_wait();
    double r = w.da[lane ^ m];
    w.bar->arrive_and_wait();
    return r;
}

// v_mfma_f32_16x16x4_f32 (cdna_hip_programming.md section 3): A[i=l&15][k=l>>4],
// B[k=l>>4][j=l&15], D[row=4*(l>>4)+r][col=l&15]; k-ordered fmaf chain.
SPX_DEVICE f32x4 mfma_16x16x4(float a, float b, f32x4 c) {
    WaveState& w = my_wave();
    int lane = ctx().tid & 63;
    w.fa[lane] = a;
    w.fb[lane] = b;
    w.bar->arrive_and_wait();
    int col = lane & 15;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (lane >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k)
            acc = std::fmaf(w.fa[k * 16 + row], w.fb[k * 16 + col], acc);
        d[r] = acc;
    }
    w.bar->arrive_and_wait();
    return d;
}

// v_mfma_f64_16x16x4_f64: same A/B lane map, D[row=(l>>4)+4*r][col=l&15]; k-ordered fma chain.
SPX_DEVICE f64x4 mfma_f64_16x16x4(double a, double b, f64x4 c) {
    WaveState& w = my_wave();
    int lane = ctx().tid & 63;
    w.da[lane] = a;
    w.db[lane] = b;
    w.bar->arrive_and_wait();
    int col = lane & 15;
    f64x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = (lane >> 4) + 4 * r;
        double acc = c[r];
        for (int k = 0; k < 4; ++k)
            acc = std::fma(w.da[k * 16 + row], w.db[k * 16 + col], acc);
        d[r] = acc;
    }
    w.bar->arrive_and_wait();
    return d;
}

// Run `body()` as a grid of `nblocks` workgroups of `nthreads` work-items,
// one workgroup at a time.
// `lds_bytes`: the dynamic LDS size the real launch passes (spx_capi.hip).  The block is
// allocated with exactly that size (rounded up to 16), so an address-sanitizer build of this
// harness (make emu-asan) reports any kernel access beyond what the GPU launch provides.
inline void launch(int64_t nblocks, int nthreads, const std::function<void()>& body,
                   size_t lds_bytes = 160 * 1024) {
    EmuState& s = emu();
    const size_t lds_alloc = (lds_bytes + 15) / 16 * 16;
    std::unique_ptr<unsigned char, void (*)(void*)> lds_block(
        static_cast<unsigned char*>(std::aligned_alloc(16, lds_alloc ? lds_alloc : 16)), std::free);
    s.lds = lds_block.get();
    s.nthreads = nthreads;
    s.nblocks = nblocks;
    int nwaves = (nthreads + 63) / 64;
    s.waves.clear();
    s.waves.resize(nwaves);
    for (int w = 0; w < nwaves; ++w) {
        int lanes = std::min(64, nthreads - 64 * w);
        s.waves[w].bar = std::make_unique<std::barrier<>>(lanes);
    }
    s.block_bar = std::make_unique<std::barrier<>>(nthreads);
    for (int64_t b = 0; b < nblocks; ++b) {
        std::memset(s.lds, 0xCD, lds_alloc);       // poison: LDS is uninitialised on a GPU
        std::vector<std::thread> th;
        th.reserve(nthreads);
        for (int t = 0; t < nthreads; ++t) {
            th.emplace_back([&, t, b]() {
                ctx().tid = t;
                ctx().bid = b;
                body();
            });
        }
        for (auto& t : th) t.join();
    }
}

}  // namespace rt
}  // namespace spx
