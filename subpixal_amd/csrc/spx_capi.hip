// libsubpixal_hip.so -- C ABI (include/subpixal_hip.h) over the gfx950 kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC spx_capi.hip
#include "spx_rt_hip.h"
#include "spx_kernels.h"
#include "spx_kernels128.h"
#include "spx_kernels32.h"
#include "spx_aux_kernels.h"
#include "spx_tables.h"
#include "../../include/subpixal_hip.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

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

struct DeviceTables {
    spx::cf* tw128 = nullptr;                 // w_128^j
    spx::cf* tw256 = nullptr;                 // w_256^j (128 tile)
    spx::cf* tw192 = nullptr;                 // w_192^j (96 tile)
    spx::cf* tw64 = nullptr;                  // w_64^j (32 tile)
    std::map<int, float*> ktab;               // upsample -> lane-major tables, 64 tile
    std::map<int, float*> ktab256;            // upsample -> lane-major tables, 128 tile
    std::map<int, float*> ktab192;            // upsample -> lane-major tables, 96 tile
    std::map<int, float*> ktab32;             // upsample -> lane-major tables, 32 tile
    int num_cu = 256;
    bool lds_attr_set = false;
};

std::mutex g_mu;
std::map<int, DeviceTables> g_dev;

int current_tables(DeviceTables** out) {
    int dev = 0;
    SPX_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceTables& t = g_dev[dev];
    if (!t.tw128) {
        std::vector<float> tw = spx::host::make_twiddles(128);
        void* p = nullptr;
        SPX_HIP(hipMalloc(&p, tw.size() * sizeof(float)));
        SPX_HIP(hipMemcpy(p, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
        t.tw128 = reinterpret_cast<spx::cf*>(p);
        std::vector<float> tw2 = spx::host::make_twiddles(256);
        SPX_HIP(hipMalloc(&p, tw2.size() * sizeof(float)));
        SPX_HIP(hipMemcpy(p, tw2.data(), tw2.size() * sizeof(float), hipMemcpyHostToDevice));
        t.tw256 = reinterpret_cast<spx::cf*>(p);
        std::vector<float> tw4 = spx::host::make_twiddles(192);
        SPX_HIP(hipMalloc(&p, tw4.size() * sizeof(float)));
        SPX_HIP(hipMemcpy(p, tw4.data(), tw4.size() * sizeof(float), hipMemcpyHostToDevice));
        t.tw192 = reinterpret_cast<spx::cf*>(p);
        std::vector<float> tw3 = spx::host::make_twiddles(64);
        SPX_HIP(hipMalloc(&p, tw3.size() * sizeof(float)));
        SPX_HIP(hipMemcpy(p, tw3.data(), tw3.size() * sizeof(float), hipMemcpyHostToDevice));
        t.tw64 = reinterpret_cast<spx::cf*>(p);
        hipDeviceProp_t prop;
        SPX_HIP(hipGetDeviceProperties(&prop, dev));
        t.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    *out = &t;
    return 0;
}

int ktab_for(DeviceTables* t, int upsample, const float** out) {
    *out = nullptr;
    const int wb = spx::host::window_blocks(upsample);
    if (wb <= 0) return 0;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = t->ktab.find(upsample);
    if (it == t->ktab.end()) {
        std::vector<float> k = spx::host::make_ktab(128, upsample, 16 * wb);
        void* p = nullptr;
        SPX_HIP(hipMalloc(&p, k.size() * sizeof(float)));
        SPX_HIP(hipMemcpy(p, k.data(), k.size() * sizeof(float), hipMemcpyHostToDevice));
        it = t->ktab.emplace(upsample, reinterpret_cast<float*>(p)).first;
    }
    *out = it->second;
    return 0;
}

int ktab256_for(DeviceTables* t, int upsample, const float** out) {
    *out = nullptr;
    const int wb = spx::host::window_blocks(upsample);
    if (wb <= 0) return 0;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = t->ktab256.find(upsample);
    if (it == t->ktab256.end()) {
        std::vector<float> k = spx::host::make_ktab256(upsample, 16 * wb);
        void* p = nullptr;
        SPX_HIP(hipMalloc(&p, k.size() * sizeof(float)));
        SPX_HIP(hipMemcpy(p, k.data(), k.size() * sizeof(float), hipMemcpyHostToDevice));
        it = t->ktab256.emplace(upsample, reinterpret_cast<float*>(p)).first;
    }
    *out = it->second;
    return 0;
}

int ktab192_for(DeviceTables* t, int upsample, const float** out) {
    *out = nullptr;
    const int wb = spx::host::window_blocks(upsample);
    if (wb <= 0) return 0;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = t->ktab192.find(upsample);
    if (it == t->ktab192.end()) {
        std::vector<float> k = spx::host::make_ktab_big(192, upsample, 16 * wb);
        void* p = nullptr;
        SPX_HIP(hipMalloc(&p, k.size() * sizeof(float)));
        SPX_HIP(hipMemcpy(p, k.data(), k.size() * sizeof(float), hipMemcpyHostToDevice));
        it = t->ktab192.emplace(upsample, reinterpret_cast<float*>(p)).first;
    }
    *out = it->second;
    return 0;
}

int ktab32_for(DeviceTables* t, int upsample, const float** out) {
    *out = nullptr;
    const int wb = spx::host::window_blocks(upsample);
    if (wb <= 0) return 0;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = t->ktab32.find(upsample);
    if (it == t->ktab32.end()) {
        std::vector<float> k = spx::host::make_ktab32(upsample, 16 * wb);
        void* p = nullptr;
        SPX_HIP(hipMalloc(&p, k.size() * sizeof(float)));
        SPX_HIP(hipMemcpy(p, k.data(), k.size() * sizeof(float), hipMemcpyHostToDevice));
        it = t->ktab32.emplace(upsample, reinterpret_cast<float*>(p)).first;
    }
    *out = it->second;
    return 0;
}

// workgroups of a 128-tile launch (each owns one 772 KiB workspace slot)
int64_t grid128(int num_cu, int64_t nbatch) {
    // SPX_GRID128_PER_CU (tuning knob, default 2): resident workgroups per CU; each owns
    // 772 KiB of workspace (at 2 per CU the total, 386 MiB, exceeds the 256 MiB Infinity Cache;
    // 1 per CU fits but measured slower: too little latency hiding)
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

template <typename K> int allow_lds(K kernel, int bytes) {
    SPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
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

template <int WB, int DBG = 0>
int launch_pair(const DeviceTables* t, const float* ref, const float* img, int64_t nbatch, int ny,
                int nx, int U, int cc_type, const float* ktab, double* out, int32_t* status,
                hipStream_t s) {
    const int lds = spx::Lds<2>::total(16 * WB);
    auto kern = spx::pair_kernel<2, WB, DBG>;
    int rc = allow_lds(kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(grid_for(t, nbatch)), dim3(spx::kThreads), lds, s, ref, img,
                       nbatch, ny, nx, U, cc_type, t->tw128, ktab, out, status);
    SPX_HIP(hipGetLastError());
    return 0;
}

// 32 tile: one wave per pair, four pairs per workgroup
template <int WB>
int launch_pair32(const DeviceTables* t, const float* ref, const float* img, int64_t nbatch, int ny,
                  int nx, int U, int cc_type, const float* ktab, double* out, int32_t* status,
                  hipStream_t s) {
    const int lds = spx::Lds32::total(16 * (WB > 0 ? WB : 1));
    auto kern = spx::pair32_kernel<WB>;
    int rc = allow_lds(kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(grid_for(t, (nbatch + 3) / 4)), dim3(spx::kThreads), lds, s, ref,
                       img, nbatch, ny, nx, U, cc_type, t->tw64, ktab, out, status);
    SPX_HIP(hipGetLastError());
    return 0;
}

// C = 4: 128 tile (period 256), C = 3: 96 tile (period 192)
template <int C, int WB, int DBG = 0>
int launch_pair128(const DeviceTables* t, const float* ref, const float* img, int64_t nbatch, int ny,
                   int nx, int U, int cc_type, const float* ktab, double* out, int32_t* status,
                   float* ws, hipStream_t s) {
    const int lds = spx::LdsBig<C>::total(16 * WB);
    auto kern = spx::pair128_kernel<C, WB, DBG>;
    int rc = allow_lds(kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid128(t->num_cu, nbatch)), dim3(spx::kThreads), lds, s,
                       ref, img, nbatch, ny, nx, U, cc_type, C == 4 ? t->tw256 : t->tw192, ktab, out,
                       status, ws);
    SPX_HIP(hipGetLastError());
    return 0;
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
    if (spx::host::window_blocks(upsample) < 0)
        return fail(SPX_E_SHAPE, "upsample must be in [1, 59]");
    DeviceTables* t = nullptr;
    int rc = current_tables(&t);
    if (rc) return rc;
    const float* k = nullptr;
    return ktab_for(t, upsample, &k);
}

size_t spx_workspace_bytes_xcorr(int64_t nbatch, int ny, int nx) {
    if (nbatch <= 0 || ny <= 0 || nx <= 0) return 0;
    if (ny <= 64 && nx <= 64) return 0;
    const size_t per_group = (ny <= 96 && nx <= 96) ? spx::kWs96Bytes : spx::kWs128Bytes;
    return (size_t)grid128(device_cus(), nbatch) * per_group;
}

// the phase-stamp diagnostic runs on the 128 tile whatever the shape
static size_t workspace_bytes_tile128(int64_t nbatch, int ny, int nx) {
    if (nbatch <= 0 || (ny <= 64 && nx <= 64)) return 0;
    return (size_t)grid128(device_cus(), nbatch) * spx::kWs128Bytes;
}

size_t spx_workspace_bytes_displacement5(int64_t nbatch, int ny, int nx, int need_icc) {
    if (nbatch <= 0 || ny <= 0 || nx <= 0) return 0;
    size_t b = spx_workspace_bytes_xcorr(nbatch, ny, nx);
    if (need_icc) b += (size_t)nbatch * 4u * (size_t)ny * (size_t)nx * sizeof(float);
    return b;
}

int spx_xcorr_refine_f32(const float* ref, const float* img, int64_t nbatch, int ny, int nx,
                         int upsample, int cc_type, double* out_dxdy, int32_t* out_status,
                         void* workspace, size_t workspace_bytes, void* stream) {
    if (nbatch < 0 || (nbatch > 0 && (!ref || !img || !out_dxdy)))
        return fail(SPX_E_ARG, "null pointer or negative batch");
    if (ny < 5 || nx < 5 || ny > SPX_MAX_SIDE || nx > SPX_MAX_SIDE)
        return fail(SPX_E_SHAPE, "pair mode supports cutouts of 5..128 pixels per side");
    const int wb = spx::host::window_blocks(upsample);
    if (wb < 0) return fail(SPX_E_SHAPE, "upsample must be in [1, 59]");
    if (nbatch == 0) return 0;
    const bool big = ny > 64 || nx > 64;            // 96 / 128 tile, FFT period 192 / 256
    if (big && (!workspace || workspace_bytes < spx_workspace_bytes_xcorr(nbatch, ny, nx)))
        return fail(SPX_E_WORKSPACE, "cutouts above 64 px need spx_workspace_bytes_xcorr() bytes");
    DeviceTables* t = nullptr;
    int rc = current_tables(&t);
    if (rc) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (big && ny <= 96 && nx <= 96) {
        const float* ktab = nullptr;
        rc = ktab192_for(t, upsample, &ktab);
        if (rc) return rc;
        float* ws = reinterpret_cast<float*>(workspace);
        switch (wb) {
        case 0: return launch_pair128<3, 0>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, ws, s);
        case 1: return launch_pair128<3, 1>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, ws, s);
        case 2: return launch_pair128<3, 2>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, ws, s);
        case 3: return launch_pair128<3, 3>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, ws, s);
        default: return launch_pair128<3, 4>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, ws, s);
        }
    }
    if (big) {
        const float* ktab = nullptr;
        rc = ktab256_for(t, upsample, &ktab);
        if (rc) return rc;
        float* ws = reinterpret_cast<float*>(workspace);
        switch (wb) {
        case 0: return launch_pair128<4, 0>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, ws, s);
        case 1: return launch_pair128<4, 1>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, ws, s);
        case 2: return launch_pair128<4, 2>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, ws, s);
        case 3: return launch_pair128<4, 3>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, ws, s);
        default: return launch_pair128<4, 4>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, ws, s);
        }
    }
    const float* ktab = nullptr;
    if (ny <= 32 && nx <= 32) {                     // 32 tile, FFT period 64
        rc = ktab32_for(t, upsample, &ktab);
        if (rc) return rc;
        switch (wb) {
        case 0: return launch_pair32<0>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, s);
        case 1: return launch_pair32<1>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, s);
        case 2: return launch_pair32<2>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, s);
        case 3: return launch_pair32<3>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, s);
        default: return launch_pair32<4>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, s);
        }
    }
    rc = ktab_for(t, upsample, &ktab);
    if (rc) return rc;
    switch (wb) {
    case 0: return launch_pair<0>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, s);
    case 1: return launch_pair<1>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, s);
    case 2: return launch_pair<2>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, s);
    case 3: return launch_pair<3>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, s);
    default: return launch_pair<4>(t, ref, img, nbatch, ny, nx, upsample, cc_type, ktab, out_dxdy, out_status, s);
    }
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
    const float* ktab = nullptr;
    rc = ktab_for(t, 10, &ktab);
    if (rc) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define SPX_PH(k) case k: return launch_pair<1, k>(t, ref, img, nbatch, ny, nx, 10, 0, ktab, out_dxdy, out_status, s);
    switch (phase) {
        SPX_PH(0) SPX_PH(1) SPX_PH(2) SPX_PH(3) SPX_PH(4) SPX_PH(5) SPX_PH(6) SPX_PH(7)
        SPX_PH(8) SPX_PH(9) SPX_PH(10) SPX_PH(11) SPX_PH(12) SPX_PH(13) SPX_PH(100)
    }
#undef SPX_PH
    return fail(SPX_E_ARG, "bad phase");
}

// same for the 128 tile at upsample 20 (full kernel with per-phase stamps only)
int spx_diag_pair128_phase(const float* ref, const float* img, int64_t nbatch, int ny, int nx,
                           double* out_dxdy, int32_t* out_status, void* workspace,
                           size_t workspace_bytes, void* stream) {
    DeviceTables* t = nullptr;
    int rc = current_tables(&t);
    if (rc) return rc;
    if (workspace_bytes < workspace_bytes_tile128(nbatch, ny, nx)) return fail(SPX_E_WORKSPACE, "workspace");
    const float* ktab = nullptr;
    rc = ktab256_for(t, 20, &ktab);
    if (rc) return rc;
    return launch_pair128<4, 2, 100>(t, ref, img, nbatch, ny, nx, 20, 0, ktab, out_dxdy, out_status,
                                  reinterpret_cast<float*>(workspace), reinterpret_cast<hipStream_t>(stream));
}
#endif

int spx_find_displacement5_f32(const float* ref, const float* im4, int64_t nbatch, int ny,
                               int nx, int cc_type, double* out_dxdy, int32_t* out_status,
                               float* out_icc, void* workspace, size_t workspace_bytes,
                               void* stream) {
    if (nbatch < 0 || (nbatch > 0 && (!ref || !im4 || !out_dxdy)))
        return fail(SPX_E_ARG, "null pointer or negative batch");
    if (ny < 3 || nx < 3 || ny > SPX_MAX_SIDE || nx > SPX_MAX_SIDE)
        return fail(SPX_E_SHAPE, "5-image mode supports cutouts of 3..128 pixels per side");
    if (nbatch == 0) return 0;
    const size_t need = spx_workspace_bytes_displacement5(nbatch, ny, nx, out_icc == nullptr);
    if (need > 0 && (!workspace || workspace_bytes < need))
        return fail(SPX_E_WORKSPACE, "workspace missing or smaller than spx_workspace_bytes_displacement5()");
    unsigned char* wsb = reinterpret_cast<unsigned char*>(workspace);
    const size_t tile_ws = spx_workspace_bytes_xcorr(nbatch, ny, nx);
    float* icc = out_icc ? out_icc : reinterpret_cast<float*>(wsb + tile_ws);
    DeviceTables* t = nullptr;
    int rc = current_tables(&t);
    if (rc) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (ny > 64 || nx > 64) {
        const int lds = spx::Lds128::total(0);
        const bool t96 = ny <= 96 && nx <= 96;          // 96 tile (period 192), else 128 tile
        auto kern = t96 ? spx::disp5_128_kernel<3> : spx::disp5_128_kernel<4>;
        rc = allow_lds(kern, lds);
        if (rc) return rc;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid128(t->num_cu, nbatch)), dim3(spx::kThreads), lds, s,
                           ref, im4, nbatch, ny, nx, cc_type, t96 ? t->tw192 : t->tw256, icc, out_dxdy,
                           out_status, reinterpret_cast<float*>(wsb));
        SPX_HIP(hipGetLastError());
        return 0;
    }
    if (ny <= 32 && nx <= 32) {
        const int lds32 = spx::Lds32::total(16);
        auto k32 = spx::disp5_32_kernel;
        rc = allow_lds(k32, lds32);
        if (rc) return rc;
        hipLaunchKernelGGL(k32, dim3(grid_for(t, (nbatch + 3) / 4)), dim3(spx::kThreads), lds32, s, ref, im4,
                           nbatch, ny, nx, cc_type, t->tw64, icc, out_dxdy, out_status);
        SPX_HIP(hipGetLastError());
        return 0;
    }
    const int lds = spx::Lds<2>::total(0);
    auto kern = spx::disp5_kernel<2>;
    rc = allow_lds(kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(grid_for(t, nbatch)), dim3(spx::kThreads), lds, s, ref, im4, nbatch,
                       ny, nx, cc_type, t->tw128, icc, out_dxdy, out_status);
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
