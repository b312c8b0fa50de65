// libsubpixal_hip.so -- C ABI (include/subpixal_hip.h) over the gfx950 kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC spx_capi.hip
#include "spx_rt_hip.h"
#include "spx_kernels.h"
#include "spx_kernels8.h"
#include "spx_kernels5.h"
#include "spx_kernels128.h"
#include "spx_kernels_big.h"
#include "spx_kernels32.h"
#include "spx_aux_kernels.h"
#include "spx_tables.h"
#include "../../include/subpixal_hip.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#ifndef SPX_DISP5_PACKED_DEFAULT
#define SPX_DISP5_PACKED_DEFAULT 1
#endif
#ifndef SPX_PAIR64_WAVES_DEFAULT
#define SPX_PAIR64_WAVES_DEFAULT 4      // measured: 25.2e6 pairs/s on 4 waves, 18.2e6 on 8 (profiles/r03)
#endif

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    return fail(SPX_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define SPX_HIP(call)                                              \
    do {                                                           \
        hipError_t e__ = (call);                                   \
        if (e__ != hipSuccess) return hip_fail(e__, #call);        \
    } while (0)

// Kernel families (spx_kernels32.h / spx_kernels.h / spx_kernels128.h) and the cutout sides
// they take.  The FFT period is the smallest the path has for which the reference's 'same'
// window (cc.py:114-126) is alias-free: P > 2n - 2 - (n-1)/2.
enum Tile { TILE32 = 0, TILE64 = 1, TILE192 = 2, NUM_TILES = 3, TILE_BIG = 3 };
constexpr int kFoldMaxSide = 85;      // 64 tile, fold path: period 128 covers cutouts up to 85 px
Tile tile_for(int ny, int nx) {
    const int n = ny > nx ? ny : nx;
    return n <= 32 ? TILE32 : (n <= kFoldMaxSide ? TILE64 : (n <= 128 ? TILE192 : TILE_BIG));
}
constexpr int kPeriod[NUM_TILES] = {64, 128, 192};

struct DeviceTables {
    bool ready = false;                          // set only after EVERY table below exists
    spx::cf* tw[NUM_TILES] = {nullptr, nullptr, nullptr};     // w_P^j
    std::map<int, float*> ktab[NUM_TILES];       // upsample -> lane-major interpolation tables
    std::map<int, float*> tw_big;                // general path: class count C -> w_{64C}^j
    std::map<std::pair<int, int>, float*> ktab_big;     // (C, upsample) -> tables
    std::set<const void*> lds_ok;                // kernels whose dynamic-LDS limit is raised
    int num_cu = 256;
    std::mutex launch_mu;                        // serialises enqueues of host threads sharing a device
};

std::mutex g_mu;                                 // guards g_dev and every DeviceTables' maps
std::map<int, DeviceTables> g_dev;

// (float64 tables travel as opaque `float*` device pointers too and are cast back at the launch)
template <typename T> int upload(const std::vector<T>& h, float** out) {
    void* p = nullptr;
    SPX_HIP(hipMalloc(&p, h.size() * sizeof(T)));
    const hipError_t e = hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return hip_fail(e, "hipMemcpy(table)");
    }
    *out = reinterpret_cast<float*>(p);
    return 0;
}

// tables of the calling thread's current device; built on first use.  Everything is built
// into locals and committed together, so a failed allocation leaves no half-initialised state.
int current_tables(DeviceTables** out) {
    int dev = 0;
    SPX_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceTables& t = g_dev[dev];
    if (!t.ready) {
        float* tw[NUM_TILES] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < NUM_TILES; ++k) {
            const int rc = upload(spx::host::make_twiddles(kPeriod[k]), &tw[k]);
            if (rc) {
                for (int j = 0; j < k; ++j) (void)hipFree(tw[j]);
                return rc;
            }
        }
        hipDeviceProp_t prop;
        const hipError_t e = hipGetDeviceProperties(&prop, dev);
        if (e != hipSuccess) {
            for (int j = 0; j < NUM_TILES; ++j) (void)hipFree(tw[j]);
            return hip_fail(e, "hipGetDeviceProperties");
        }
        for (int k = 0; k < NUM_TILES; ++k) t.tw[k] = reinterpret_cast<spx::cf*>(tw[k]);
        t.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        t.ready = true;
    }
    *out = &t;
    return 0;
}

// Lock order everywhere: g_mu, then a device's launch_mu (spx_shutdown's order).  A launch looks its
// tables up under g_mu, takes launch_mu while still holding it and only then lets g_mu go (TableLock):
// spx_shutdown on another thread therefore cannot free a table between the lookup and the enqueue.
struct TableLock {
    std::unique_lock<std::mutex> g;
    std::unique_lock<std::mutex> launch;
    TableLock() : g(g_mu) {}
    void enter_launch(DeviceTables* t) {
        launch = std::unique_lock<std::mutex>(t->launch_mu);
        g.unlock();
    }
};

// interpolation tables of one tile family for `upsample` (nullptr for upsample 1); caller holds g_mu.
// `f64_64`: the 64 tile's tables as float64 in the float64 MFMA's result order (make_ktab_f64), for the four-wave
// kernel's float64-refine form (PairArgs::refine_f64); the eight-wave A/B kernel always reads float32 ones.
int ktab_for(DeviceTables* t, Tile tile, int upsample, const float** out, bool f64_64 = false) {
    *out = nullptr;
    const int wb = spx::host::window_blocks(upsample);
    if (wb <= 0) return 0;
    f64_64 = f64_64 && tile == TILE64;
    const int key = upsample + (f64_64 ? (1 << 20) : 0);
    auto it = t->ktab[tile].find(key);
    if (it == t->ktab[tile].end()) {
        float* p = nullptr;
        // period 192: float64 tables (its refine stage accumulates in float64)
        const int rc = tile == TILE32 ? upload(spx::host::make_ktab32(upsample, 16 * wb), &p)
                     : tile == TILE64 ? (f64_64 ? upload(spx::host::make_ktab_f64(128, upsample, 16 * wb), &p)
                                                : upload(spx::host::make_ktab(128, upsample, 16 * wb), &p))
                                      : upload(spx::host::make_ktab_big_f64(192, upsample, 16 * wb), &p);
        if (rc) return rc;
        it = t->ktab[tile].emplace(key, p).first;
    }
    *out = it->second;
    return 0;
}
// both forms of a pair-mode launch's tables: `ktab` for the kernel family's refine as this call wants it
// (`refine_f64`: the 64 tile's float64 form), `ktab_f32` for the eight-wave 64-tile kernel (the same pointer
// unless `ktab` is the float64 table)
int pair_tables_for(DeviceTables* t, Tile tile, int upsample, bool refine_f64, const float** ktab,
                    const float** ktab_f32) {
    int rc = ktab_for(t, tile, upsample, ktab, refine_f64);
    if (rc) return rc;
    if (tile == TILE64 && refine_f64) return ktab_for(t, tile, upsample, ktab_f32, false);
    *ktab_f32 = *ktab;
    return 0;
}
// what a call's `refine` argument means for the 64 tile (the other families have one form each: float32 on the
// 32 tile, float64 above 85 px).  The default is float32 at every upsample.  A rule "float64 from three window
// blocks on" was in for an hour of round 3 -- float32 loses 2 % / 10 % of the pairs of sigma 11..15 px spots at
// upsample 39 / 59, float64 none (profiles/r03/width_precision_256.txt) -- until its cost was measured: the
// float64 form takes 12.0 instead of 6.2 ms per 1e5 pairs at upsample 28..43 and 26.3 instead of 11.9 at 59
// (profiles/r03/default_rule_cost.txt: 336 / 512 float64 MFMAs per wave instead of 80; since brought to 8.8 and
// 19.7 ms, f64_live3_ab.txt / f64_live4_ab.txt).  Taking 29..40 % off every such call's rate to cover that corner is
// the caller's decision, not a default: SPX_REFINE_F64.
int refine64_is_f64(int refine, int wb, bool* f64) {
    (void)wb;
    if (refine != SPX_REFINE_DEFAULT && refine != SPX_REFINE_F64 && refine != SPX_REFINE_F32)
        return fail(SPX_E_ARG, "refine must be SPX_REFINE_DEFAULT, SPX_REFINE_F64 or SPX_REFINE_F32");
    *f64 = refine == SPX_REFINE_F64 || (refine == SPX_REFINE_DEFAULT && spx::kRefine64DefaultF64);
    return 0;
}

// general path (cutouts above 128 px): twiddles and tables per class count, built on first use;
// caller holds g_mu
int big_tables_for(DeviceTables* t, int C, int upsample, const spx::cf** tw, const float** ktab) {
    *tw = nullptr;
    *ktab = nullptr;
    auto it = t->tw_big.find(C);
    if (it == t->tw_big.end()) {
        float* p = nullptr;
        const int rc = upload(spx::host::make_twiddles(64 * C), &p);
        if (rc) return rc;
        it = t->tw_big.emplace(C, p).first;
    }
    *tw = reinterpret_cast<const spx::cf*>(it->second);
    const int wb = spx::host::window_blocks(upsample);
    if (wb > 0) {
        auto kt = t->ktab_big.find({C, upsample});
        if (kt == t->ktab_big.end()) {
            float* p = nullptr;
            const int rc = upload(spx::host::make_ktab_big_f64(64 * C, upsample, 16 * wb), &p);
            if (rc) return rc;
            kt = t->ktab_big.emplace(std::make_pair(C, upsample), p).first;
        }
        *ktab = kt->second;
    }
    return 0;
}
// two resident workgroups per CU (as the period-192 path) up to C = 8, one above: the workspace
// of a workgroup is megabytes there (4 C^2 planes + the P x P convolution: 21 MB at C = 16)
int64_t grid_general(int num_cu, int64_t nbatch, int C) {
    const int64_t cap = (int64_t)num_cu * (C <= 8 ? 2 : 1);
    return nbatch < cap ? nbatch : cap;
}

// workgroups of a period-192 launch (each owns one workspace slot)
int64_t grid_big(int num_cu, int64_t nbatch) {
    // SPX_GRID128_PER_CU (tuning knob, default 2): resident workgroups per CU
    static const int per_cu = [] {
        const char* e = getenv("SPX_GRID128_PER_CU");
        const int v = e ? atoi(e) : 2;
        return v >= 1 && v <= 4 ? v : 2;
    }();
    const int64_t cap = (int64_t)num_cu * per_cu;
    return nbatch < cap ? nbatch : cap;
}
int device_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        return 256;
    return n;
}

// raise a kernel's dynamic-LDS limit: once per kernel instance and device (spx_prepare() does
// it for every instance a later launch can pick, so a captured launch issues no such call)
// (callers hold t->launch_mu)
template <typename K> int allow_lds(DeviceTables* t, K kernel, int bytes) {
    const void* key = reinterpret_cast<const void*>(kernel);
    if (t->lds_ok.count(key)) return 0;
    SPX_HIP(hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    t->lds_ok.insert(key);
    return 0;
}

// workgroups per launch: enough to fill 256 CUs x 2 resident groups several times
// over; the kernels grid-stride over the batch.
unsigned grid_for(const DeviceTables* t, int64_t nbatch) {
    // SPX_GRID_PER_CU (tuning knob): workgroups launched per CU (2 are resident at a time)
    static const int per_cu = [] {
        const char* e = getenv("SPX_GRID_PER_CU");
        const int v = e ? atoi(e) : 32;
        return v >= 1 && v <= 1024 ? v : 32;
    }();
    const int64_t cap = (int64_t)t->num_cu * per_cu;
    return (unsigned)(nbatch < cap ? nbatch : cap);
}

struct PairArgs {
    int64_t nbatch;
    int ny, nx, U, cc_type;
    const float* ktab;
    const float* ktab_f32;       // for the eight-wave 64-tile kernel (pair_tables_for)
    bool refine_f64;             // 64 tile: the float64-refine form of the four-wave kernel (`ktab` is then float64)
    double* out;
    int32_t* status;
    float* ws;
    hipStream_t s;
};

// One pair-mode kernel instance: `launch` false only raises its LDS limit (spx_prepare).
template <int WB, bool FOLD, typename TIn, int DBG, typename R>
int run_pair64_as(DeviceTables* t, const TIn* ref, const TIn* img, const PairArgs& a, bool launch) {
    const int lds = spx::Lds<2>::total(16 * WB);
    auto kern = spx::pair_kernel<2, WB, DBG, FOLD, TIn, R>;
    int rc = allow_lds(t, kern, lds);
    if (rc || !launch) return rc;
    hipLaunchKernelGGL(kern, dim3(grid_for(t, a.nbatch)), dim3(spx::kThreads), lds, a.s, ref, img,
                       a.nbatch, a.ny, a.nx, a.U, a.cc_type, t->tw[TILE64], a.ktab, a.out, a.status);
    SPX_HIP(hipGetLastError());
    return 0;
}
// The float64-refine form exists for the product instances with a refine stage (DBG = 0, WB > 0); without
// `launch` (spx_prepare) BOTH forms get their LDS limit raised, so either may be launched inside a capture.
template <int WB, bool FOLD, typename TIn, int DBG = 0>
int run_pair64(DeviceTables* t, const TIn* ref, const TIn* img, const PairArgs& a, bool launch) {
    if constexpr (DBG == 0 && WB > 0) {
        if (!launch) {
            const int rc = run_pair64_as<WB, FOLD, TIn, DBG, spx::RefineF64>(t, ref, img, a, false);
            if (rc) return rc;
        } else if (a.refine_f64) {
            return run_pair64_as<WB, FOLD, TIn, DBG, spx::RefineF64>(t, ref, img, a, true);
        }
    }
    return run_pair64_as<WB, FOLD, TIn, DBG, spx::RefineF32>(t, ref, img, a, launch);
}
// the same tile on eight waves per pair (spx_kernels8.h; cutouts up to 64 px, no fold path)
template <int WB, typename TIn, int DBG = 0>
int run_pair64_w8(DeviceTables* t, const TIn* ref, const TIn* img, const PairArgs& a, bool launch) {
    const int lds = spx::w8::L8::total(16 * WB);
    auto kern = spx::w8::pair8_kernel<WB, DBG, TIn>;
    int rc = allow_lds(t, kern, lds);
    if (rc || !launch) return rc;
    hipLaunchKernelGGL(kern, dim3(grid_for(t, a.nbatch)), dim3(spx::w8::kT8), lds, a.s, ref, img,
                       a.nbatch, a.ny, a.nx, a.U, a.cc_type, t->tw[TILE64], a.ktab_f32, a.out, a.status);
    SPX_HIP(hipGetLastError());
    return 0;
}
// SPX_PAIR64_WAVES = 4 | 8 (A/B knob; read once): which kernel takes pair-mode cutouts of 33..64 px
int pair64_waves() {
    static const int w = [] {
        const char* e = getenv("SPX_PAIR64_WAVES");
        const int v = e ? atoi(e) : SPX_PAIR64_WAVES_DEFAULT;
        return v == 4 ? 4 : 8;
    }();
    return w;
}
// 32 tile: one wave per pair, four pairs per workgroup
template <int WB, typename TIn>
int run_pair32(DeviceTables* t, const TIn* ref, const TIn* img, const PairArgs& a, bool launch) {
    const int lds = spx::Lds32::total(16 * (WB > 0 ? WB : 1));
    auto kern = spx::pair32_kernel<WB, TIn>;
    int rc = allow_lds(t, kern, lds);
    if (rc || !launch) return rc;
    hipLaunchKernelGGL(kern, dim3(grid_for(t, (a.nbatch + 3) / 4)), dim3(spx::kThreads), lds, a.s, ref,
                       img, a.nbatch, a.ny, a.nx, a.U, a.cc_type, t->tw[TILE32], a.ktab, a.out, a.status);
    SPX_HIP(hipGetLastError());
    return 0;
}
// period 192 (9 spectral classes, per-workgroup workspace)
template <int WB, typename TIn, int DBG = 0>
int run_pair192(DeviceTables* t, const TIn* ref, const TIn* img, const PairArgs& a, bool launch) {
    const int lds = spx::LdsBig<3>::total(16 * WB);
    auto kern = spx::pair128_kernel<3, WB, DBG, TIn>;
    int rc = allow_lds(t, kern, lds);
    if (rc || !launch) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid_big(t->num_cu, a.nbatch)), dim3(spx::kThreads), lds, a.s,
                       ref, img, a.nbatch, a.ny, a.nx, a.U, a.cc_type, t->tw[TILE192],
                       reinterpret_cast<const double*>(a.ktab), a.out, a.status, a.ws);
    SPX_HIP(hipGetLastError());
    return 0;
}

template <int WB, typename TIn>
int run_pair_wb(DeviceTables* t, Tile tile, bool fold, const TIn* ref, const TIn* img,
                const PairArgs& a, bool launch) {
    switch (tile) {
    case TILE32: return run_pair32<WB, TIn>(t, ref, img, a, launch);
    case TILE64:
        if (fold) return run_pair64<WB, true, TIn>(t, ref, img, a, launch);
        if constexpr (sizeof(TIn) == 4) {      // the eight-wave A/B kernel is built for float32 cutouts only
            if (!launch) {          // spx_prepare: both variants stay launchable inside a capture
                const int rc = run_pair64_w8<WB, TIn>(t, ref, img, a, false);
                return rc ? rc : run_pair64<WB, false, TIn>(t, ref, img, a, false);
            }
            if (pair64_waves() == 8 && !a.refine_f64) return run_pair64_w8<WB, TIn>(t, ref, img, a, true);
        }
        return run_pair64<WB, false, TIn>(t, ref, img, a, launch);
    default: return run_pair192<WB, TIn>(t, ref, img, a, launch);
    }
}
template <typename TIn>
int run_pair(DeviceTables* t, int wb, Tile tile, bool fold, const TIn* ref, const TIn* img,
             const PairArgs& a, bool launch) {
    switch (wb) {
    case 0: return run_pair_wb<0, TIn>(t, tile, fold, ref, img, a, launch);
    case 1: return run_pair_wb<1, TIn>(t, tile, fold, ref, img, a, launch);
    case 2: return run_pair_wb<2, TIn>(t, tile, fold, ref, img, a, launch);
    case 3: return run_pair_wb<3, TIn>(t, tile, fold, ref, img, a, launch);
    default: return run_pair_wb<4, TIn>(t, tile, fold, ref, img, a, launch);
    }
}

struct Disp5Args {
    int64_t nbatch;
    int ny, nx, cc_type;
    float* icc;
    double* out;
    int32_t* status;
    float* ws;
    hipStream_t s;
    spx::ItemTable items;        // per-item offsets and shapes, or nulls for a uniform batch
};
template <typename TIn>
int run_disp5(DeviceTables* t, Tile tile, bool fold, const TIn* ref, const TIn* im4,
              const Disp5Args& a, bool launch) {
    if (tile == TILE192) {
        const int lds = spx::LdsBig<3>::total(0);
        auto kern = spx::disp5_128_kernel<3, TIn>;
        int rc = allow_lds(t, kern, lds);
        if (rc || !launch) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid_big(t->num_cu, a.nbatch)), dim3(spx::kThreads), lds,
                           a.s, ref, im4, a.nbatch, a.ny, a.nx, a.cc_type, t->tw[TILE192], a.icc, a.out,
                           a.status, a.ws, a.items);
        SPX_HIP(hipGetLastError());
        return 0;
    }
    if (tile == TILE32) {
        const int lds = spx::Lds32::total(16);
        auto kern = spx::disp5_32_kernel<TIn>;
        int rc = allow_lds(t, kern, lds);
        if (rc || !launch) return rc;
        hipLaunchKernelGGL(kern, dim3(grid_for(t, (a.nbatch + 3) / 4)), dim3(spx::kThreads), lds, a.s, ref,
                           im4, a.nbatch, a.ny, a.nx, a.cc_type, t->tw[TILE32], a.icc, a.out, a.status, a.items);
        SPX_HIP(hipGetLastError());
        return 0;
    }
    // 64 tile: the packed kernel (five transforms per source, one workgroup per CU: spx_kernels5.h) or the
    // round-2 one (eight, two per CU).  Measured on one box, alternating (profiles/r03/disp5_rstd_and_packed_ab.txt,
    // disp5_packed_ab_first.txt): up to 64 px packed is faster for plain CC (+10 %) and NCC (+7 %) and level for
    // ZNCC (-1 %); on the fold path (65..85 px: a staged region twice the size, uncovered on a CU with one
    // workgroup) it is 7-15 % slower.  It takes the cutouts up to 64 px.
    // SPX_DISP5_PACKED = 0 never | 1 that rule (default) | 2 always  (A/B knob, read once)
    static const int mode = [] {
        const char* e = getenv("SPX_DISP5_PACKED");
        const int v = e ? atoi(e) : SPX_DISP5_PACKED_DEFAULT;
        return v < 0 || v > 2 ? 1 : v;
    }();
    const bool packed = mode == 2 || (mode == 1 && !fold);
    {
        const int lds5 = fold ? spx::p5::L5<true>::TOTAL : spx::p5::L5<false>::TOTAL;
        auto k5 = fold ? spx::p5::disp5p_kernel<true, TIn> : spx::p5::disp5p_kernel<false, TIn>;
        int rc5 = allow_lds(t, k5, lds5);
        if (rc5) return rc5;
        if (launch && packed) {
            hipLaunchKernelGGL(k5, dim3(grid_for(t, a.nbatch)), dim3(spx::kThreads), lds5, a.s, ref, im4, a.nbatch,
                               a.ny, a.nx, a.cc_type, t->tw[TILE64], a.icc, a.out, a.status, a.items);
            SPX_HIP(hipGetLastError());
            return 0;
        }
    }
    const int lds = spx::Lds<2>::total(0);
    auto kern = fold ? spx::disp5_kernel<2, true, TIn> : spx::disp5_kernel<2, false, TIn>;
    int rc = allow_lds(t, kern, lds);
    if (rc || !launch) return rc;
    hipLaunchKernelGGL(kern, dim3(grid_for(t, a.nbatch)), dim3(spx::kThreads), lds, a.s, ref, im4, a.nbatch,
                       a.ny, a.nx, a.cc_type, t->tw[TILE64], a.icc, a.out, a.status, a.items);
    SPX_HIP(hipGetLastError());
    return 0;
}

size_t ws_bytes_xcorr(int64_t nbatch, int ny, int nx) {
    if (nbatch <= 0 || ny <= 0 || nx <= 0) return 0;
    const Tile tile = tile_for(ny, nx);
    if (tile == TILE_BIG) {
        const int C = spx::big_class_count(ny, nx);
        return (size_t)grid_general(device_cus(), nbatch, C) * spx::big_ws_floats(C) * sizeof(float);
    }
    if (tile != TILE192) return 0;
    return (size_t)grid_big(device_cus(), nbatch) * spx::kWs96Bytes;
}

template <int WB, typename TIn>
int run_pair_general(DeviceTables* t, const TIn* ref, const TIn* img, const PairArgs& a, int C,
                     const spx::cf* tw, bool launch) {
    const int lds = spx::LdsGen::total(16 * WB);
    auto kern = spx::pair_big_kernel<WB, TIn>;
    int rc = allow_lds(t, kern, lds);
    if (rc || !launch) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid_general(t->num_cu, a.nbatch, C)), dim3(spx::kThreads), lds,
                       a.s, ref, img, a.nbatch, a.ny, a.nx, a.U, a.cc_type, C, tw,
                       reinterpret_cast<const double*>(a.ktab), a.out, a.status, a.ws);
    SPX_HIP(hipGetLastError());
    return 0;
}
template <typename TIn>
int run_pair_general_wb(DeviceTables* t, int wb, const TIn* ref, const TIn* img, const PairArgs& a,
                        int C, const spx::cf* tw, bool launch) {
    switch (wb) {
    case 0: return run_pair_general<0, TIn>(t, ref, img, a, C, tw, launch);
    case 1: return run_pair_general<1, TIn>(t, ref, img, a, C, tw, launch);
    case 2: return run_pair_general<2, TIn>(t, ref, img, a, C, tw, launch);
    case 3: return run_pair_general<3, TIn>(t, ref, img, a, C, tw, launch);
    default: return run_pair_general<4, TIn>(t, ref, img, a, C, tw, launch);
    }
}
template <typename TIn>
int run_disp5_general(DeviceTables* t, const TIn* ref, const TIn* im4, const Disp5Args& a, int C,
                      const spx::cf* tw, bool launch) {
    const int lds = spx::LdsGen::total(0);
    auto kern = spx::disp5_big_kernel<TIn>;
    int rc = allow_lds(t, kern, lds);
    if (rc || !launch) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid_general(t->num_cu, a.nbatch, C)), dim3(spx::kThreads), lds,
                       a.s, ref, im4, a.nbatch, a.ny, a.nx, a.cc_type, C, tw, a.icc, a.out, a.status, a.ws);
    SPX_HIP(hipGetLastError());
    return 0;
}

template <typename TIn>
int xcorr_refine(const TIn* ref, const TIn* img, int64_t nbatch, int ny, int nx, int upsample,
                 int cc_type, int refine, double* out_dxdy, int32_t* out_status, void* workspace,
                 size_t workspace_bytes, void* stream) {
    if (nbatch < 0 || (nbatch > 0 && (!ref || !img || !out_dxdy)))
        return fail(SPX_E_ARG, "null pointer or negative batch");
    bool refine_f64 = false;
    if (int rr = refine64_is_f64(refine, spx::host::window_blocks(upsample), &refine_f64)) return rr;
    if (ny < 5 || nx < 5 || ny > SPX_MAX_SIDE || nx > SPX_MAX_SIDE)
        return fail(SPX_E_SHAPE, "pair mode supports cutouts of 5..682 pixels per side");
    const int wb = spx::host::window_blocks(upsample);
    if (wb < 0) return fail(SPX_E_SHAPE, "upsample must be in [1, 59]");
    if (nbatch == 0) return 0;
    const Tile tile = tile_for(ny, nx);
    if (tile >= TILE192 && (!workspace || workspace_bytes < ws_bytes_xcorr(nbatch, ny, nx)))
        return fail(SPX_E_WORKSPACE, "cutouts above 85 px need spx_workspace_bytes_xcorr() bytes");
    DeviceTables* t = nullptr;
    int rc = current_tables(&t);
    if (rc) return rc;
    PairArgs a = {};
    a.nbatch = nbatch; a.ny = ny; a.nx = nx; a.U = upsample; a.cc_type = cc_type;
    a.out = out_dxdy; a.status = out_status;
    a.ws = reinterpret_cast<float*>(workspace);
    a.s = reinterpret_cast<hipStream_t>(stream);
    TableLock lk;
    if (!t->ready) return fail(SPX_E_ARG, "spx_shutdown() ran between the call's start and its launch");
    if (tile == TILE_BIG) {            // general path: class count and tables at run time
        // Above 128 px the float32 transforms limit the refinement at very fine grids: 11..25-px-wide spots
        // at upsample >= 40 reached 1.2e-3 px against the float64 definition (profiles/r02/sweeps_late.txt),
        // past the 1e-3 px this library promises -- refused rather than returned out of tolerance.
        if (upsample > SPX_MAX_UPSAMPLE_GENERAL)
            return fail(SPX_E_SHAPE, "cutouts above 128 px take upsample 1..39");
        const int C = spx::big_class_count(ny, nx);
        const spx::cf* tw = nullptr;
        rc = big_tables_for(t, C, upsample, &tw, &a.ktab);
        if (rc) return rc;
        lk.enter_launch(t);
        return run_pair_general_wb<TIn>(t, wb, ref, img, a, C, tw, true);
    }
    // the eight-wave A/B kernel has no float64 form: a float64-refine call takes the four-wave kernel
    a.refine_f64 = refine_f64 && tile == TILE64;
    rc = pair_tables_for(t, tile, upsample, a.refine_f64, &a.ktab, &a.ktab_f32);
    if (rc) return rc;
    lk.enter_launch(t);
    return run_pair<TIn>(t, wb, tile, ny > 64 || nx > 64, ref, img, a, true);
}

template <typename TIn>
int find_displacement5(const TIn* ref, const TIn* im4, int64_t nbatch, int ny, int nx, int cc_type,
                       double* out_dxdy, int32_t* out_status, float* out_icc, void* workspace,
                       size_t workspace_bytes, void* stream) {
    if (nbatch < 0 || (nbatch > 0 && (!ref || !im4 || !out_dxdy)))
        return fail(SPX_E_ARG, "null pointer or negative batch");
    if (ny < 3 || nx < 3 || ny > SPX_MAX_SIDE || nx > SPX_MAX_SIDE)
        return fail(SPX_E_SHAPE, "5-image mode supports cutouts of 3..682 pixels per side");
    if (nbatch == 0) return 0;
    const size_t need = spx_workspace_bytes_displacement5(nbatch, ny, nx, out_icc == nullptr);
    if (need > 0 && (!workspace || workspace_bytes < need))
        return fail(SPX_E_WORKSPACE, "workspace missing or smaller than spx_workspace_bytes_displacement5()");
    unsigned char* wsb = reinterpret_cast<unsigned char*>(workspace);
    const size_t tile_ws = ws_bytes_xcorr(nbatch, ny, nx);
    DeviceTables* t = nullptr;
    int rc = current_tables(&t);
    if (rc) return rc;
    Disp5Args a;
    a.items = spx::ItemTable{nullptr, nullptr, 0, 0};
    a.nbatch = nbatch; a.ny = ny; a.nx = nx; a.cc_type = cc_type;
    a.icc = out_icc ? out_icc : reinterpret_cast<float*>(wsb + tile_ws);
    a.out = out_dxdy; a.status = out_status;
    a.ws = reinterpret_cast<float*>(wsb);
    a.s = reinterpret_cast<hipStream_t>(stream);
    TableLock lk;
    if (!t->ready) return fail(SPX_E_ARG, "spx_shutdown() ran between the call's start and its launch");
    if (tile_for(ny, nx) == TILE_BIG) {
        const int C = spx::big_class_count(ny, nx);
        const spx::cf* tw = nullptr;
        const float* unused = nullptr;
        rc = big_tables_for(t, C, 1, &tw, &unused);
        if (rc) return rc;
        lk.enter_launch(t);
        return run_disp5_general<TIn>(t, ref, im4, a, C, tw, true);
    }
    lk.enter_launch(t);
    return run_disp5<TIn>(t, tile_for(ny, nx), ny > 64 || nx > 64, ref, im4, a, true);
}

// reference mode over cutouts of different shapes, all within one kernel family (`family_side`)
template <typename TIn>
int find_displacement5_var(const TIn* ref, const TIn* im4, const int64_t* item_offset,
                           const int32_t* item_shape, int64_t nbatch, int family_side, int cc_type,
                           double* out_dxdy, int32_t* out_status, float* out_icc, void* workspace,
                           size_t workspace_bytes, void* stream) {
    if (nbatch < 0 || (nbatch > 0 && (!ref || !im4 || !item_offset || !item_shape || !out_dxdy || !out_icc)))
        return fail(SPX_E_ARG, "null pointer or negative batch");
    if (family_side < 3 || family_side > 128)
        return fail(SPX_E_SHAPE, "variable-shape batches take cutouts of 3..128 pixels per side");
    if (nbatch == 0) return 0;
    const Tile tile = tile_for(family_side, family_side);
    const size_t need = ws_bytes_xcorr(nbatch, family_side, family_side);
    if (need > 0 && (!workspace || workspace_bytes < need))
        return fail(SPX_E_WORKSPACE, "workspace missing or smaller than spx_workspace_bytes_xcorr()");
    DeviceTables* t = nullptr;
    int rc = current_tables(&t);
    if (rc) return rc;
    const bool fold = family_side > 64;
    Disp5Args a;
    // the largest side the family launched here takes: 32 / 64 / 85 (fold path) / 128
    a.items = spx::ItemTable{item_offset, item_shape,
                             tile == TILE32 ? 32 : (tile == TILE64 ? (fold ? kFoldMaxSide : 64) : 128), 0};
    a.nbatch = nbatch; a.ny = family_side; a.nx = family_side; a.cc_type = cc_type;
    a.icc = out_icc;
    a.out = out_dxdy; a.status = out_status;
    a.ws = reinterpret_cast<float*>(workspace);
    a.s = reinterpret_cast<hipStream_t>(stream);
    TableLock lk;
    if (!t->ready) return fail(SPX_E_ARG, "spx_shutdown() ran between the call's start and its launch");
    lk.enter_launch(t);
    return run_disp5<TIn>(t, tile, fold, ref, im4, a, true);
}

}  // namespace

extern "C" {

int spx_abi_version(void) { return SPX_ABI_VERSION; }

const char* spx_last_error(void) { return g_err.c_str(); }

int spx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int spx_init(int device) {
    SPX_HIP(hipSetDevice(device));
    DeviceTables* t = nullptr;
    return current_tables(&t);
}

int spx_prepare(int upsample) {
    const int wb = spx::host::window_blocks(upsample);
    if (wb < 0) return fail(SPX_E_SHAPE, "upsample must be in [1, 59]");
    DeviceTables* t = nullptr;
    int rc = current_tables(&t);
    if (rc) return rc;
    PairArgs pa = {};
    Disp5Args da = {};
    {
        std::lock_guard<std::mutex> gl(g_mu);
        for (int k = 0; k < NUM_TILES; ++k) {
            const float* kt = nullptr;
            const float* kt32 = nullptr;        // both refine forms of the 64 tile: no allocation in a later call
            if ((rc = pair_tables_for(t, (Tile)k, upsample, false, &kt, &kt32))) return rc;
            if ((Tile)k == TILE64 && (rc = pair_tables_for(t, TILE64, upsample, true, &kt, &kt32))) return rc;
        }
    }
    for (int k = 0; k < NUM_TILES; ++k) {
        std::lock_guard<std::mutex> lk(t->launch_mu);
        const Tile tile = (Tile)k;
        for (int fold = 0; fold < (tile == TILE64 ? 2 : 1); ++fold) {
            if ((rc = run_pair<float>(t, wb, tile, fold != 0, nullptr, nullptr, pa, false))) return rc;
            if ((rc = run_pair<double>(t, wb, tile, fold != 0, nullptr, nullptr, pa, false))) return rc;
            if ((rc = run_disp5<float>(t, tile, fold != 0, nullptr, nullptr, da, false))) return rc;
            if ((rc = run_disp5<double>(t, tile, fold != 0, nullptr, nullptr, da, false))) return rc;
        }
    }
    {   // general path (cutouts above 128 px): its tables depend on the cutout size and are built on
        // first use (spx_prepare_shape() builds them up front); the LDS limits are raised here
        std::lock_guard<std::mutex> lk(t->launch_mu);
        if ((rc = run_pair_general_wb<float>(t, wb, nullptr, nullptr, pa, 4, nullptr, false))) return rc;
        if ((rc = run_pair_general_wb<double>(t, wb, nullptr, nullptr, pa, 4, nullptr, false))) return rc;
        if ((rc = run_disp5_general<float>(t, nullptr, nullptr, da, 4, nullptr, false))) return rc;
        if ((rc = run_disp5_general<double>(t, nullptr, nullptr, da, 4, nullptr, false))) return rc;
    }
    return 0;
}

int spx_prepare_shape(int ny, int nx, int upsample) {
    int rc = spx_prepare(upsample);
    if (rc) return rc;
    if (ny < 3 || nx < 3 || ny > SPX_MAX_SIDE || nx > SPX_MAX_SIDE)
        return fail(SPX_E_SHAPE, "cutouts of 3..682 pixels per side");
    if (tile_for(ny, nx) != TILE_BIG) return 0;
    DeviceTables* t = nullptr;
    if ((rc = current_tables(&t))) return rc;
    const spx::cf* tw = nullptr;
    const float* kt = nullptr;
    std::lock_guard<std::mutex> gl(g_mu);
    return big_tables_for(t, spx::big_class_count(ny, nx), upsample, &tw, &kt);
}

int spx_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    int prev = 0;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    int rc = 0;
    for (auto& kv : g_dev) {
        DeviceTables& t = kv.second;
        std::lock_guard<std::mutex> ll(t.launch_mu);
        if (hipSetDevice(kv.first) != hipSuccess) { rc = SPX_E_HIP; continue; }
        if (hipDeviceSynchronize() != hipSuccess) rc = SPX_E_HIP;      // no launch may still read a table
        for (int k = 0; k < NUM_TILES; ++k) {
            if (t.tw[k]) (void)hipFree(t.tw[k]);
            t.tw[k] = nullptr;
            for (auto& e : t.ktab[k]) (void)hipFree(e.second);
            t.ktab[k].clear();
        }
        for (auto& e : t.tw_big) (void)hipFree(e.second);
        t.tw_big.clear();
        for (auto& e : t.ktab_big) (void)hipFree(e.second);
        t.ktab_big.clear();
        t.lds_ok.clear();
        t.ready = false;
    }
    if (have_prev) (void)hipSetDevice(prev);
    if (rc) return fail(rc, "spx_shutdown: a device could not be synchronised");
    return 0;
}

size_t spx_workspace_bytes_xcorr(int64_t nbatch, int ny, int nx) { return ws_bytes_xcorr(nbatch, ny, nx); }

// the phase-stamp diagnostic runs on the period-192 path whatever the shape
static size_t workspace_bytes_big(int64_t nbatch) {
    if (nbatch <= 0) return 0;
    return (size_t)grid_big(device_cus(), nbatch) * spx::kWs96Bytes;
}

size_t spx_workspace_bytes_displacement5(int64_t nbatch, int ny, int nx, int need_icc) {
    if (nbatch <= 0 || ny <= 0 || nx <= 0) return 0;
    size_t b = ws_bytes_xcorr(nbatch, ny, nx);
    if (need_icc) b += (size_t)nbatch * 4u * (size_t)ny * (size_t)nx * sizeof(float);
    return b;
}

int spx_xcorr_refine_f32(const float* ref, const float* img, int64_t nbatch, int ny, int nx,
                         int upsample, int cc_type, double* out_dxdy, int32_t* out_status,
                         void* workspace, size_t workspace_bytes, void* stream) {
    return xcorr_refine<float>(ref, img, nbatch, ny, nx, upsample, cc_type, SPX_REFINE_DEFAULT, out_dxdy,
                               out_status, workspace, workspace_bytes, stream);
}
int spx_xcorr_refine_ex_f32(const float* ref, const float* img, int64_t nbatch, int ny, int nx,
                            int upsample, int cc_type, int refine, double* out_dxdy, int32_t* out_status,
                            void* workspace, size_t workspace_bytes, void* stream) {
    return xcorr_refine<float>(ref, img, nbatch, ny, nx, upsample, cc_type, refine, out_dxdy, out_status,
                               workspace, workspace_bytes, stream);
}
int spx_xcorr_refine_f64(const double* ref, const double* img, int64_t nbatch, int ny, int nx,
                         int upsample, int cc_type, double* out_dxdy, int32_t* out_status,
                         void* workspace, size_t workspace_bytes, void* stream) {
    return xcorr_refine<double>(ref, img, nbatch, ny, nx, upsample, cc_type, SPX_REFINE_DEFAULT, out_dxdy,
                                out_status, workspace, workspace_bytes, stream);
}
int spx_xcorr_refine_ex_f64(const double* ref, const double* img, int64_t nbatch, int ny, int nx,
                            int upsample, int cc_type, int refine, double* out_dxdy, int32_t* out_status,
                            void* workspace, size_t workspace_bytes, void* stream) {
    return xcorr_refine<double>(ref, img, nbatch, ny, nx, upsample, cc_type, refine, out_dxdy, out_status,
                                workspace, workspace_bytes, stream);
}

#ifdef SPX_PHASE_TIMING
// Diagnostic library only (`make diag`): the pair kernel (upsample 10) cut short after
// phase `phase` (1..13, see tools/phase_timing.py); results are invalid.  phase 100 =
// full kernel with per-phase cycle stamps: out_status must have room for nbatch int32
// + 20 uint64 (8-byte aligned: nbatch even), zeroed by the caller.
int spx_diag_pair_phase(const float* ref, const float* img, int64_t nbatch, int ny, int nx,
                        int phase, double* out_dxdy, int32_t* out_status, void* stream) {
    DeviceTables* t = nullptr;
    int rc = current_tables(&t);
    if (rc) return rc;
    PairArgs a = {};
    a.nbatch = nbatch; a.ny = ny; a.nx = nx; a.U = 10; a.cc_type = 0; a.out = out_dxdy;
    a.status = out_status; a.ws = nullptr; a.s = reinterpret_cast<hipStream_t>(stream);
    {
        std::lock_guard<std::mutex> gl(g_mu);
        a.refine_f64 = false;
        rc = pair_tables_for(t, TILE64, 10, false, &a.ktab, &a.ktab_f32);
    }
    if (rc) return rc;
    const bool fold = ny > 64 || nx > 64;
#define SPX_PH(k) case k: return fold ? run_pair64<1, true, float, k>(t, ref, img, a, true) \
                                      : run_pair64<1, false, float, k>(t, ref, img, a, true);
    // phases 200 + k: the eight-wave kernel (k = 1 staging only, 10 through the planes, 100 stamps)
    if (phase == 201) return run_pair64_w8<1, float, 1>(t, ref, img, a, true);
    if (phase == 210) return run_pair64_w8<1, float, 10>(t, ref, img, a, true);
    if (phase == 300) return run_pair64_w8<1, float, 100>(t, ref, img, a, true);
    switch (phase) {
        SPX_PH(0) SPX_PH(1) SPX_PH(2) SPX_PH(3) SPX_PH(4) SPX_PH(5) SPX_PH(6) SPX_PH(7)
        SPX_PH(8) SPX_PH(9) SPX_PH(10) SPX_PH(11) SPX_PH(12) SPX_PH(13) SPX_PH(100)
    }
#undef SPX_PH
    return fail(SPX_E_ARG, "bad phase");
}

// same for the period-192 path at upsample 20 (full kernel with per-phase stamps only)
int spx_diag_pair128_phase(const float* ref, const float* img, int64_t nbatch, int ny, int nx,
                           double* out_dxdy, int32_t* out_status, void* workspace,
                           size_t workspace_bytes, void* stream) {
    DeviceTables* t = nullptr;
    int rc = current_tables(&t);
    if (rc) return rc;
    if (workspace_bytes < workspace_bytes_big(nbatch)) return fail(SPX_E_WORKSPACE, "workspace");
    PairArgs a = {};
    a.nbatch = nbatch; a.ny = ny; a.nx = nx; a.U = 20; a.cc_type = 0; a.out = out_dxdy;
    a.status = out_status; a.ws = reinterpret_cast<float*>(workspace);
    a.s = reinterpret_cast<hipStream_t>(stream);
    {
        std::lock_guard<std::mutex> gl(g_mu);
        rc = ktab_for(t, TILE192, 20, &a.ktab);
    }
    if (rc) return rc;
    return run_pair192<2, float, 100>(t, ref, img, a, true);
}
size_t spx_diag_workspace_bytes_big(int64_t nbatch) { return workspace_bytes_big(nbatch); }
#endif

int spx_find_displacement5_f32(const float* ref, const float* im4, int64_t nbatch, int ny,
                               int nx, int cc_type, double* out_dxdy, int32_t* out_status,
                               float* out_icc, void* workspace, size_t workspace_bytes,
                               void* stream) {
    return find_displacement5<float>(ref, im4, nbatch, ny, nx, cc_type, out_dxdy, out_status, out_icc,
                                     workspace, workspace_bytes, stream);
}
int spx_find_displacement5_f64(const double* ref, const double* im4, int64_t nbatch, int ny,
                               int nx, int cc_type, double* out_dxdy, int32_t* out_status,
                               float* out_icc, void* workspace, size_t workspace_bytes,
                               void* stream) {
    return find_displacement5<double>(ref, im4, nbatch, ny, nx, cc_type, out_dxdy, out_status, out_icc,
                                      workspace, workspace_bytes, stream);
}

int spx_find_displacement5_var_f32(const float* ref, const float* im4, const int64_t* item_offset,
                                   const int32_t* item_shape, int64_t nbatch, int family_side,
                                   int cc_type, double* out_dxdy, int32_t* out_status, float* out_icc,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    return find_displacement5_var<float>(ref, im4, item_offset, item_shape, nbatch, family_side, cc_type,
                                         out_dxdy, out_status, out_icc, workspace, workspace_bytes, stream);
}
int spx_find_displacement5_var_f64(const double* ref, const double* im4, const int64_t* item_offset,
                                   const int32_t* item_shape, int64_t nbatch, int family_side,
                                   int cc_type, double* out_dxdy, int32_t* out_status, float* out_icc,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    return find_displacement5_var<double>(ref, im4, item_offset, item_shape, nbatch, family_side, cc_type,
                                          out_dxdy, out_status, out_icc, workspace, workspace_bytes, stream);
}

int spx_find_displacement5_catalog_f32(const float* ref, const float* im4, const int64_t* item_offset,
                                       const int32_t* item_shape, int64_t nbatch, int family_mask,
                                       int cc_type, double* out_dxdy, int32_t* out_status,
                                       float* out_icc, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    if (nbatch < 0 || (nbatch > 0 && (!ref || !im4 || !item_offset || !item_shape || !out_dxdy || !out_icc)))
        return fail(SPX_E_ARG, "null pointer or negative batch");
    if (family_mask & ~15) return fail(SPX_E_ARG, "family_mask has bits 0..3");
    if (nbatch == 0 || family_mask == 0) return 0;
    const size_t need = (family_mask & SPX_FAMILY_128) ? ws_bytes_xcorr(nbatch, 128, 128) : 0;
    if (need > 0 && (!workspace || workspace_bytes < need))
        return fail(SPX_E_WORKSPACE, "workspace missing or smaller than spx_workspace_bytes_xcorr(nbatch, 128, 128)");
    DeviceTables* t = nullptr;
    int rc = current_tables(&t);
    if (rc) return rc;
    // (largest side taken, sides at or below this are another family's, tile, fold path)
    static const struct { int bit, max_side, skip_below; Tile tile; bool fold; } fam[4] = {
        {SPX_FAMILY_32, 32, 2, TILE32, false}, {SPX_FAMILY_64, 64, 32, TILE64, false},
        {SPX_FAMILY_85, kFoldMaxSide, 64, TILE64, true}, {SPX_FAMILY_128, 128, kFoldMaxSide, TILE192, false}};
    TableLock lk;
    if (!t->ready) return fail(SPX_E_ARG, "spx_shutdown() ran between the call's start and its launch");
    lk.enter_launch(t);
    for (int k = 0; k < 4; ++k) {
        if (!(family_mask & fam[k].bit)) continue;
        Disp5Args a;
        a.items = spx::ItemTable{item_offset, item_shape, fam[k].max_side, fam[k].skip_below};
        a.nbatch = nbatch; a.ny = fam[k].max_side; a.nx = fam[k].max_side; a.cc_type = cc_type;
        a.icc = out_icc;
        a.out = out_dxdy; a.status = out_status;
        a.ws = reinterpret_cast<float*>(workspace);
        a.s = reinterpret_cast<hipStream_t>(stream);
        if ((rc = run_disp5<float>(t, fam[k].tile, fam[k].fold, ref, im4, a, true))) return rc;
    }
    return 0;
}

int spx_gather_cutouts_var_f32(const float* frame, const uint8_t* fmask, int fny, int fnx,
                               const int32_t* boxes, int64_t nbatch, const int64_t* item_offset,
                               float fill, float* packed, const int32_t* seg, const int32_t* ids,
                               void* stream) {
    if (nbatch < 0 || (nbatch > 0 && (!frame || !boxes || !item_offset || !packed)))
        return fail(SPX_E_ARG, "null pointer or negative batch");
    if ((seg == nullptr) != (ids == nullptr))
        return fail(SPX_E_ARG, "seg and ids must be given together");
    if (fny < 1 || fnx < 1) return fail(SPX_E_SHAPE, "bad shape");
    if (nbatch == 0) return 0;
    const unsigned grid = (unsigned)(nbatch < 65535 * 16 ? nbatch : 65535 * 16);
    hipLaunchKernelGGL(spx::gather_cutouts_var_kernel, dim3(grid), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), frame, fmask, fny, fnx, boxes, nbatch,
                       item_offset, fill, packed, seg, ids);
    SPX_HIP(hipGetLastError());
    return 0;
}

int spx_blot4_var_f32(const float* src, const int64_t* src_offset, const int32_t* src_shape,
                      int64_t nbatch, const double* map, int degree, const float* gain,
                      const int64_t* dst_offset, const int32_t* dst_shape, float* im4, void* stream) {
    if (nbatch < 0 || (nbatch > 0 && (!src || !src_offset || !src_shape || !map || !dst_offset || !dst_shape || !im4)))
        return fail(SPX_E_ARG, "null pointer or negative batch");
    if (degree < 0 || degree > 5) return fail(SPX_E_ARG, "degree: 0 (affine) or 1..5 (polynomial)");
    if (nbatch == 0) return 0;
    const unsigned grid = (unsigned)(nbatch < 65535 * 16 ? nbatch : 65535 * 16);
    hipLaunchKernelGGL(spx::blot4_var_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       src, src_offset, src_shape, nbatch, map, degree, gain, dst_offset, dst_shape, im4);
    SPX_HIP(hipGetLastError());
    return 0;
}

int spx_find_peak_f64(const double* image, const uint8_t* mask, const double* guess,
                      int64_t nbatch, int ny, int nx, int fit_wx, int fit_wy, int search_wx,
                      int search_wy, double* out_xy, int32_t* out_status, void* stream) {
    if (nbatch < 0 || (nbatch > 0 && (!image || !out_xy)))
        return fail(SPX_E_ARG, "null pointer or negative batch");
    if (ny < 1 || nx < 1 || (int64_t)ny * nx > (int64_t)1 << 30)
        return fail(SPX_E_SHAPE, "bad image shape");
    if (fit_wx < 1 || fit_wy < 1 || search_wx < 0 || search_wy < 0 ||
        ((search_wx == 0) != (search_wy == 0)))
        return fail(SPX_E_ARG, "box dimensions must be positive (search box: both 0 = off)");
    if ((int64_t)fit_wx * fit_wy > spx::kPeakMaxFitPoints)
        return fail(SPX_E_SHAPE, "fit box larger than 1024 points");
    if (nbatch == 0) return 0;
    const unsigned grid = (unsigned)(nbatch < 65535 * 16 ? nbatch : 65535 * 16);
    hipLaunchKernelGGL(spx::find_peak_kernel, dim3(grid), dim3(spx::kThreads), 0,
                       reinterpret_cast<hipStream_t>(stream), image, mask, guess, nbatch, ny, nx,
                       fit_wx, fit_wy, search_wx, search_wy, out_xy, out_status);
    SPX_HIP(hipGetLastError());
    return 0;
}

int spx_gather_cutouts_f32(const float* frame, const uint8_t* fmask, int fny, int fnx,
                           const int32_t* boxes, int64_t nbatch, int tny, int tnx, float fill,
                           float* tiles, const int32_t* seg, const int32_t* ids, void* stream) {
    if (nbatch < 0 || (nbatch > 0 && (!frame || !boxes || !tiles)))
        return fail(SPX_E_ARG, "null pointer or negative batch");
    if ((seg == nullptr) != (ids == nullptr))
        return fail(SPX_E_ARG, "seg and ids must be given together");
    if (fny < 1 || fnx < 1 || tny < 1 || tnx < 1) return fail(SPX_E_SHAPE, "bad shape");
    if (nbatch == 0) return 0;
    const int64_t total = nbatch * tny * tnx;
    const int64_t blocks = (total + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 65535 * 32 ? blocks : 65535 * 32);
    hipLaunchKernelGGL(spx::gather_cutouts_kernel, dim3(grid), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), frame, fmask, fny, fnx, boxes, nbatch,
                       tny, tnx, fill, tiles, seg, ids);
    SPX_HIP(hipGetLastError());
    return 0;
}

int spx_label_bboxes_i32(const int32_t* seg, int fny, int fnx, int32_t max_label,
                         int32_t* boxes, int32_t* counts, void* stream) {
    if (!seg || !boxes || !counts) return fail(SPX_E_ARG, "null pointer");
    if (fny < 1 || fnx < 1 || max_label < 0) return fail(SPX_E_SHAPE, "bad shape");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int nl = max_label + 1;
    hipLaunchKernelGGL(spx::label_bbox_init_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, s,
                       boxes, counts, nl);
    SPX_HIP(hipGetLastError());
    const int64_t total = (int64_t)fny * ((fnx + 3) / 4);
    const int64_t blocks = (total + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(spx::label_bbox_kernel, dim3(grid), dim3(256), 0, s, seg, fny, fnx,
                       (int)max_label, boxes, counts);
    SPX_HIP(hipGetLastError());
    return 0;
}

int spx_blot_affine4_f32(const float* src, int64_t nbatch, int sny, int snx, const double* affine,
                         const float* gain, int ny, int nx, float* im4, void* stream) {
    if (nbatch < 0 || (nbatch > 0 && (!src || !affine || !im4)))
        return fail(SPX_E_ARG, "null pointer or negative batch");
    if (sny < 6 || snx < 6 || sny > 4096 || snx > 4096 || ny < 1 || nx < 1 || ny > 4096 || nx > 4096)
        return fail(SPX_E_SHAPE, "source tiles must be 6..4096 px per side, targets 1..4096");
    if (nbatch == 0) return 0;
    const int64_t total = nbatch * 4 * ny * nx;
    const int64_t blocks = (total + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 65536 ? blocks : 65536);
    hipLaunchKernelGGL(spx::blot_affine4_kernel, dim3(grid), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), src, nbatch, sny, snx, affine, gain, ny,
                       nx, im4);
    SPX_HIP(hipGetLastError());
    return 0;
}

int spx_blot_poly4_f32(const float* src, int64_t nbatch, int sny, int snx, const double* coef,
                       int degree, const float* gain, int ny, int nx, float* im4, void* stream) {
    if (nbatch < 0 || (nbatch > 0 && (!src || !coef || !im4)))
        return fail(SPX_E_ARG, "null pointer or negative batch");
    if (degree < 1 || degree > 5) return fail(SPX_E_ARG, "polynomial degree must be 1..5");
    if (sny < 6 || snx < 6 || sny > 4096 || snx > 4096 || ny < 1 || nx < 1 || ny > 4096 || nx > 4096)
        return fail(SPX_E_SHAPE, "source tiles must be 6..4096 px per side, targets 1..4096");
    if (nbatch == 0) return 0;
    const int64_t total = nbatch * 4 * ny * nx;
    const int64_t blocks = (total + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 65536 ? blocks : 65536);
    hipLaunchKernelGGL(spx::blot_poly4_kernel, dim3(grid), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), src, nbatch, sny, snx, coef, degree, gain,
                       ny, nx, im4);
    SPX_HIP(hipGetLastError());
    return 0;
}

int spx_gen_gaussian_pairs_f32(uint64_t seed, int64_t first_index, int64_t nbatch, int n,
                               float sigma_lo, float sigma_hi, float max_shift, float* ref,
                               float* img, double* truth_dxdy, void* stream) {
    if (nbatch < 0 || (nbatch > 0 && (!ref || !img)))
        return fail(SPX_E_ARG, "null pointer or negative batch");
    if (n < 1 || n > 4096) return fail(SPX_E_SHAPE, "bad tile size");
    if (nbatch == 0) return 0;
    const unsigned grid = (unsigned)(nbatch < 65535 * 16 ? nbatch : 65535 * 16);
    hipLaunchKernelGGL(spx::gen_pairs_kernel, dim3(grid), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), seed, first_index, nbatch, n, sigma_lo,
                       sigma_hi, max_shift, ref, img, truth_dxdy);
    SPX_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
