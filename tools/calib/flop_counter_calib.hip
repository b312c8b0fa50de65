// Calibration of the SQ_INSTS_VALU_{FMA,ADD,MUL}_F32 counters for PACKED f32 instructions (tools/gpu_r3_evidence.sh):
// kernel `scalar_fma` issues N v_fma_f32 per wave, `packed_fma` N v_pk_fma_f32, `packed_add` N v_pk_add_f32,
// `packed_mul` N v_pk_mul_f32.  If a packed instruction counts once, the packed kernels report the scalar
// kernel's count; if it counts per operation, twice that.  Build: hipcc --offload-arch=gfx950 -O3 -o calib.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int N = 4096;
__global__ void scalar_fma(float* p) {
    float a = p[threadIdx.x], b = 1.0001f, c = 0.5f;
#pragma unroll 64
    for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    p[threadIdx.x] = a;
}
__global__ void packed_fma(float* p) {
    f32x2 a = {p[threadIdx.x], p[threadIdx.x + 64]}, b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
#pragma unroll 64
    for (int i = 0; i < N; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    p[threadIdx.x] = a.x + a.y;
}
__global__ void packed_add(float* p) {
    f32x2 a = {p[threadIdx.x], p[threadIdx.x + 64]}, b = {1.0e-3f, 2.0e-3f};
#pragma unroll 64
    for (int i = 0; i < N; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    p[threadIdx.x] = a.x + a.y;
}
__global__ void packed_mul(float* p) {
    f32x2 a = {p[threadIdx.x], p[threadIdx.x + 64]}, b = {1.0001f, 0.9999f};
#pragma unroll 64
    for (int i = 0; i < N; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
    p[threadIdx.x] = a.x + a.y;
}
int main() {
    float* d;
    if (hipMalloc(&d, 4096) != hipSuccess) return 1;
    hipMemset(d, 0, 4096);
    const int blocks = 1024;              // 1024 waves of 64 lanes per kernel
    hipLaunchKernelGGL(scalar_fma, dim3(blocks), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(packed_fma, dim3(blocks), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(packed_add, dim3(blocks), dim3(64), 0, 0, d);
    hipLaunchKernelGGL(packed_mul, dim3(blocks), dim3(64), 0, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("issued per kernel: %d instructions x %d waves = %lld wave-instructions\n", N, blocks, (long long)N * blocks);
    return 0;
}
