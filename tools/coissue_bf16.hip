// Micro-benchmark: inside ONE wave, do independent VALU instructions (the three-plane split of the next item) issue in the shadow
// of v_mfma_f32_16x16x32_bf16 (the products of the current item)?  And how does that change with 1 / 2 / 3 waves per SIMD?
// Per iteration: 24 MFMAs in four accumulator chains (the x3 product pattern of one 32-pair item) and/or 8 split3_pair groups
// (104 VALU instructions) on loop-carried values.
// build: hipcc --offload-arch=gfx950 -O3 -I include -I unidet3d_amd/csrc tools/coissue_bf16.hip -o tools/bin/coissue_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "u3d_common.h"
using namespace u3d;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    extern __shared__ float pad[];
    const int lane = threadIdx.x & 63;
    f32x4 t00 = {0, 0, 0, 0}, t01 = t00, t10 = t00, t11 = t00;
    f32x4 va = {lane * 1e-3f, lane * 2e-3f, 1.f, 2.f}, vb = {lane * 3e-3f, 0.5f, 0.25f, 3.f};
    bf16x8 w[3], x[3];
    split3_x8(va, vb, w);
    split3_x8(vb, va, x);
    f32x4 ra = va, rb = vb, rc = va * 1.5f, rd = vb * 0.75f;       // "gathered rows" of the next item (loop-carried)
    for (int i = 0; i < iters; ++i) {
        bf16x8 n0[3], n1[3];
        if constexpr (MODE & 2) {
            split3_x8(ra, rb, n0);
            split3_x8(rc, rd, n1);
        }
        if constexpr (MODE & 1) {
#pragma unroll
            for (int o = 2; o >= 0; --o)
#pragma unroll
                for (int qa = 0; qa <= o; ++qa) {
                    t00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[qa], x[o - qa], t00, 0, 0, 0);
                    t01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[qa], x[(o - qa + 1) % 3], t01, 0, 0, 0);
                    t10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[(qa + 1) % 3], x[o - qa], t10, 0, 0, 0);
                    t11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[(qa + 2) % 3], x[o - qa], t11, 0, 0, 0);
                }
        }
        if constexpr (MODE & 2) {       // next iteration multiplies what this one split; the rows change a little every iteration
            x[0] = n0[0]; x[1] = n0[1]; x[2] = n1[2];
            ra = ra * 1.0001f + __builtin_bit_cast(f32x4, n1[0]) * 1e-30f; rb += 0.5f; rc = rc * 0.9999f; rd -= 0.25f;
        }
        if constexpr (MODE & 4) {
#pragma unroll
            for (int u = 0; u < 24; ++u) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0); }
        }
    }
    f32x4 r = t00 + t01 + t10 + t11 + ra + rb + rc + rd;
    out[blockIdx.x * 256 + threadIdx.x] = r[0] + r[1] + r[2] + r[3] + (float)x[0][0] + pad[0] * 0.f;
}

template <int MODE>
static void run(const char* name, float* out, int wps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const size_t lds = wps == 1 ? 100 * 1024 : (wps == 2 ? 60 * 1024 : (wps == 3 ? 45 * 1024 : 30 * 1024));      // workgroups (= waves per SIMD) per CU through the LDS footprint
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * wps), dim3(256), lds, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-44s waves/SIMD %d: %8.3f ms = %7.1f cycles / iteration / SIMD (2.4 GHz)\n", name, wps, ms, ms * 1e-3 * 2.4e9 / iters / wps);
}

int main() {
    float* out; hipMalloc(&out, 2048 * 256 * 4);
    for (int wps = 1; wps <= 4; ++wps) {
        run<1>("24 MFMA", out, wps);
        run<2>("2 x split3_x8 (104 VALU)", out, wps);
        run<3>("both, compiler order", out, wps);
        run<7>("both, sched_group_barrier 1 MFMA : 5 VALU", out, wps);
    }
    return 0;
}
