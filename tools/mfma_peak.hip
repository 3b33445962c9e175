// Micro-benchmark: sustained rate of the fp32 MFMA instructions on gfx950 (register operands only).
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int CHAINS>
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
    f32x4 d[CHAINS];
    for (int c = 0; c < CHAINS; ++c) d[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) d[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += d[c][0] + d[c][1] + d[c][2] + d[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
    f32x16 d[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) d[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) d[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, d[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += d[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
static void run(const char* name, F launch, double flops) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 2000;
    for (int wgs : {256, 512, 1024, 2048}) {
        const double f16 = 2.0 * 16 * 16 * 4 * 8.0 * iters * wgs * 4, f32 = 2.0 * 32 * 32 * 2 * 8.0 * iters * wgs * 4;
        char nm[64];
        snprintf(nm, 64, "16x16x4 chains=1 wgs=%d", wgs); run(nm, [&] { hipLaunchKernelGGL(k16<1>, dim3(wgs), dim3(256), 0, 0, out, iters); }, f16);
        snprintf(nm, 64, "16x16x4 chains=2 wgs=%d", wgs); run(nm, [&] { hipLaunchKernelGGL(k16<2>, dim3(wgs), dim3(256), 0, 0, out, iters); }, f16 * 2);
        snprintf(nm, 64, "16x16x4 chains=4 wgs=%d", wgs); run(nm, [&] { hipLaunchKernelGGL(k16<4>, dim3(wgs), dim3(256), 0, 0, out, iters); }, f16 * 4);
        snprintf(nm, 64, "32x32x2 chains=1 wgs=%d", wgs); run(nm, [&] { hipLaunchKernelGGL(k32<1>, dim3(wgs), dim3(256), 0, 0, out, iters); }, f32);
        snprintf(nm, 64, "32x32x2 chains=2 wgs=%d", wgs); run(nm, [&] { hipLaunchKernelGGL(k32<2>, dim3(wgs), dim3(256), 0, 0, out, iters); }, f32 * 2);
    }
    return 0;
}
