// Micro-benchmark: do VALU / LDS / VMEM instructions of one wave overlap with MFMAs of ANOTHER wave on the same SIMD?
// 8 waves per workgroup (2 per SIMD): waves 0-3 run an MFMA loop, waves 4-7 run the "other" loop.
// build: hipcc --offload-arch=gfx950 -O3 tools/coissue.hip -o tools/bin/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;

// mode bit0: MFMA waves active; other: 0 none, 1 VALU fp32 fma, 2 VALU 64-bit mad (address math), 3 LDS read-modify-write, 4 ds_bpermute
__global__ __launch_bounds__(512) void k(float* out, int iters, int do_mfma, int other) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float res = 0.f;
    if (wave < 4) {
        if (do_mfma) {
            f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
            float a = lane * 1e-3f, b = lane * 2e-3f;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, d1, 0, 0, 0);
                }
            }
            res = d0[0] + d1[1];
        }
    } else if (other == 1) {
        float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 64; ++u) { x0 = x0 * 1.0001f + 0.5f; x1 = x1 * 1.0001f + 0.5f; x2 = x2 * 1.0001f + 0.5f; x3 = x3 * 1.0001f + 0.5f; }
        }
        res = x0 + x1 + x2 + x3;
    } else if (other == 2) {
        unsigned long long x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
        const unsigned long long m = 0x100000001ull + blockIdx.x;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) { x0 = x0 * m + 3; x1 = x1 * m + 5; x2 = x2 * m + 7; x3 = x3 * m + 9; }
        }
        res = (float)(x0 + x1 + x2 + x3);
    } else if (other == 3) {
        float* p = lds + (wave - 4) * 2048 + lane;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) p[u * 64] += 1.0f;
        }
        res = p[0];
    } else if (other == 4) {
        int v = lane;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) v = __shfl(v, (v + 1) & 63, 64);
        }
        res = v;
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

int main() {
    float* out; hipMalloc(&out, 2048 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, wgs = 256 * 4;
    const char* names[] = {"none", "VALU fp32 fma x256/iter", "VALU u64 mad x64/iter", "LDS rmw x16/iter", "ds_bpermute x16/iter"};
    for (int other = 0; other < 5; ++other)
        for (int do_mfma = 0; do_mfma < 2; ++do_mfma) {
            if (!other && !do_mfma) continue;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(wgs), dim3(512), 0, 0, out, iters, do_mfma, other);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("mfma=%d other=%-26s %8.3f ms\n", do_mfma, names[other], ms);
        }
    return 0;
}
