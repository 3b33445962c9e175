// Micro-benchmark + exactness check of the three-plane split (u3d_common.h split3_pair) against forms built on gfx950's packed
// fp32 -> bf16 conversion:
//   old : h = (bits + 0x8000) & 0xffff0000 (round half up on the bit pattern), m and l by truncation; 13 VALU per pair
//   cvt : every level rounds to nearest even with v_cvt_pk_bf16_f32, the two remainders of a pair come from one v_pk_add_f32
// Each thread splits a stream of values (loop-carried so nothing hoists), planes are XORed into an accumulator; reports cycles per
// pair-split per wave, and checks h + m + l == x exactly (in fp64) on a value sweep.
// build: hipcc --offload-arch=gfx950 -O3 tools/split_bench.hip -o tools/bin/split_bench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ void split_old(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned ah = (__builtin_bit_cast(unsigned, a) + 0x8000u) & 0xffff0000u, bh = (__builtin_bit_cast(unsigned, b) + 0x8000u) & 0xffff0000u;
    const float a1 = a - __builtin_bit_cast(float, ah), b1 = b - __builtin_bit_cast(float, bh);
    const unsigned a1b = __builtin_bit_cast(unsigned, a1), b1b = __builtin_bit_cast(unsigned, b1);
    const float a2 = a1 - __builtin_bit_cast(float, a1b & 0xffff0000u), b2 = b1 - __builtin_bit_cast(float, b1b & 0xffff0000u);
    h = __builtin_amdgcn_perm(bh, ah, 0x07060302u);
    m = __builtin_amdgcn_perm(b1b, a1b, 0x07060302u);
    l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b2), __builtin_bit_cast(unsigned, a2), 0x07060302u);
}
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split_cvt(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk(a, b);
    f32x2 r = f32x2{a, b} - f32x2{__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
    m = cvt_pk(r[0], r[1]);
    r -= f32x2{__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xffff0000u)};
    l = cvt_pk(r[0], r[1]);
}

template <int MODE>
__global__ __launch_bounds__(256) void bench(unsigned* out, int iters, unsigned long long* cyc) {
    float a = threadIdx.x * 1.000123f + 0.37f, b = threadIdx.x * -0.77f + 1.3f, c = a * 0.5f, d = b * 3.f;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        unsigned h0, m0, l0, h1, m1, l1;
        if (MODE == 0) { split_old(a, b, h0, m0, l0); split_old(c, d, h1, m1, l1); }
        else { split_cvt(a, b, h0, m0, l0); split_cvt(c, d, h1, m1, l1); }
        acc ^= h0 ^ m0 ^ l0 ^ h1 ^ m1 ^ l1;
        a = a * 1.0001f + 0.25f; b = b * 0.9999f - 0.125f; c += 1.5f; d -= 0.75f;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
__global__ void check(const float* x, int n, unsigned* planes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned h, m, l;
    if (MODE == 0) split_old(x[2 * i], x[2 * i + 1], h, m, l); else split_cvt(x[2 * i], x[2 * i + 1], h, m, l);
    planes[3 * i] = h; planes[3 * i + 1] = m; planes[3 * i + 2] = l;
}

static float bf(unsigned w, int hi) { unsigned b = hi ? (w & 0xffff0000u) : (w << 16); float f; memcpy(&f, &b, 4); return f; }

int main() {
    unsigned* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode)
        for (int wgs : {1, 2, 4}) {             // 1 / 2 / 4 waves per SIMD on one CU-sized grid
            unsigned long long c = 0;
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(bench<0>, dim3(256 * wgs), dim3(256), 0, 0, out, iters, cyc);
                else hipLaunchKernelGGL(bench<1>, dim3(256 * wgs), dim3(256), 0, 0, out, iters, cyc);
                hipDeviceSynchronize();
            }
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%s  %d wave(s)/SIMD: %.1f s_memtime ticks per pair-split per wave (x 2.4 GHz / 100 MHz = %.1f core cycles)\n", mode ? "cvt" : "old", wgs,
                   (double)c / iters / 2, (double)c / iters / 2 * 24.0);
        }
    // exactness
    const int n = 1 << 20;
    float* hx = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) {
        unsigned bits = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 30);
        if (((bits >> 23) & 0xff) == 0xff) bits &= 0x7f7fffffu | 0x80000000u;          // no inf / nan
        if (((bits >> 23) & 0xff) < 20) bits |= 20u << 23;                               // keep the third plane normal
        memcpy(&hx[i], &bits, 4);
    }
    hx[0] = 1.f; hx[1] = 1.f + ldexpf(1.f, -23); hx[2] = 255.f / 256.f; hx[3] = 1.99999988f; hx[4] = 1.00390625f; hx[5] = 1.005859375f; hx[6] = 0.f; hx[7] = -0.f;
    float* dx; unsigned* dp; hipMalloc(&dx, n * 4); hipMalloc(&dp, (size_t)n / 2 * 12);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    unsigned* hp = (unsigned*)malloc((size_t)n / 2 * 12);
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(check<0>, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, dp);
        else hipLaunchKernelGGL(check<1>, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, dp);
        hipMemcpy(hp, dp, (size_t)n / 2 * 12, hipMemcpyDeviceToHost);
        long bad = 0; double pos1 = 0, cnt1 = 0, pos2 = 0, cnt2 = 0, maxr1 = 0, maxr2 = 0;
        for (int i = 0; i < n; ++i) {
            const unsigned* w = hp + 3 * (i / 2);
            const double h = bf(w[0], i & 1), m = bf(w[1], i & 1), l = bf(w[2], i & 1), x = hx[i];
            if (h + m + l != x) ++bad;
            if (x != 0) {
                const double r1 = (x - h) / x, r2 = m != 0 ? (x - h - m) / m : 0;
                if (r1 != 0) { pos1 += r1 > 0; ++cnt1; }
                if (r2 != 0) { pos2 += r2 > 0; ++cnt2; }
                if (fabs(r1) > maxr1) maxr1 = fabs(r1);
                if (fabs(r2) > maxr2) maxr2 = fabs(r2);
            }
        }
        printf("%s  exact: %ld of %d values differ from h + m + l;  remainder after h: max %.3g of x, same sign as x in %.3f;  after m: max %.3g of m, same sign as m in %.3f\n",
               mode ? "cvt" : "old", bad, n, maxr1, pos1 / cnt1, maxr2, pos2 / cnt2);
    }
    return 0;
}
