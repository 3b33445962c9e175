// K9 BatchNorm(+ReLU) over voxel rows, training mode, on gfx950.  HBM-bound: every pass streams
// [n, C] fp32 with 16-byte coalesced loads; statistics are accumulated in fp64 so that
// var = E[x^2] - mean^2 is as accurate as torch's two-pass kernel, and so that the per-rank partial
// sums can be all-reduced across data-parallel ranks (SyncBatchNorm) between stats and apply.
// Replaces nn.SyncBatchNorm / nn.BatchNorm1d(eps=1e-4, momentum=0.1) + nn.ReLU at
// unidet3d/spconv_unet.py:42,49,119-124,147,177 and unidet3d/unidet3d.py:104-111.
#include <stdlib.h>

#include "u3d_common.h"

namespace u3d {

// What follows a statistics kernel once every workgroup has written its partial row (fp64 [2C]) to the workspace: sum the rows of
// each column in a fixed order (deterministic), store sums[0..2C) (+ the row count) and, for the forward pass, finalize the layer
// (mean / invstd / scale / shift, running statistics).
struct BnFin {
    double* sums;            // [2C+1]
    double rows;             // row count of this rank
    int set_rows;            // store rows into sums[2C]
    const float* gamma;      // finalize when st != nullptr
    const float* beta;
    float eps, momentum;
    float* running_mean;
    float* running_var;
    float* st;               // mean, invstd, scale, shift [4][C]
    int64_t* nbt;
};

// channels c = wave0, wave0 + nw, ...: one wave per channel, lanes over the partial rows, fixed shuffle tree
__device__ __forceinline__ void bn_finish_body(const double* partial, int nblk, int C, const BnFin& f, int wave0, int nw, bool first) {
    const int lane = threadIdx.x & 63;
    for (int c = wave0; c < C; c += nw) {
        double t1 = 0.0, t2 = 0.0;
        for (int b = lane; b < nblk; b += 64) {
            t1 += partial[(int64_t)b * 2 * C + c];
            t2 += partial[(int64_t)b * 2 * C + C + c];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { t1 += __shfl_xor(t1, d, 64); t2 += __shfl_xor(t2, d, 64); }
        if (lane) continue;
        f.sums[c] = t1;
        f.sums[C + c] = t2;
        if (!f.st) continue;
        const double m = t1 / f.rows;
        double var = t2 / f.rows - m * m;
        if (var < 0.0) var = 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)f.eps));
        const float mf = (float)m;
        f.st[c] = mf;
        f.st[C + c] = is;
        const float sc = f.gamma[c] * is;
        f.st[2 * C + c] = sc;
        f.st[3 * C + c] = f.beta[c] - mf * sc;
        if (f.running_mean) {
            const double unbiased = f.rows > 1.0 ? var * f.rows / (f.rows - 1.0) : var;
            f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * mf;
            f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unbiased;
        }
    }
    if (first && threadIdx.x == 0) {
        if (f.set_rows) f.sums[2 * C] = f.rows;
        if (f.nbt) *f.nbt += 1;
    }
}

// Second launch of a statistics pass: sums the per-workgroup partial rows in a fixed order (deterministic) and, for the forward pass,
// finalizes the layer; 4 channels per workgroup.  (Round 3 also tried folding this into the statistics kernel with a threadfence
// reduction -- the last workgroup to arrive sums: one launch less per pass, but the agent-scope fences made the statistics kernels
// 3-6x slower on the 8-XCD part, with full or with release- / acquire-only fences)
__global__ __launch_bounds__(256) void bn_sum_k(const double* __restrict__ partial, int nblk, int C, BnFin f) {
    bn_finish_body(partial, nblk, C, f, blockIdx.x * 4 + (threadIdx.x >> 6), gridDim.x * 4, blockIdx.x == 0);
}

// mode 0: sums = [sum x, sum x^2]; mode 1: backward sums [sum dy', sum dy'*xhat]
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                   const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                   int64_t n, int C, double* sums /* partials [gridDim.x][2C] */) {
    __shared__ double sh[256 * 8];
    const int lpr = C >> 2;
    const int rpb = 256 / lpr;
    const int tid = threadIdx.x;
    const int c4 = tid % lpr, slot = tid / lpr;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (slot < rpb) {
        float4 mu = make_float4(0, 0, 0, 0), is = mu, sc = mu, sf = mu;
        if (MODE == 1) {
            mu = *reinterpret_cast<const float4*>(mean + c4 * 4);
            is = *reinterpret_cast<const float4*>(invstd + c4 * 4);
            sc = *reinterpret_cast<const float4*>(scale + c4 * 4);
            sf = *reinterpret_cast<const float4*>(shift + c4 * 4);
        }
        // 4 rows per trip: four independent 16-byte loads in flight per lane hide the HBM latency
        const int64_t stride = (int64_t)gridDim.x * rpb;
        for (int64_t r0 = (int64_t)blockIdx.x * rpb + slot; r0 < n; r0 += 4 * stride) {
            float4 v4[4], g4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t r = r0 + u * stride;
                v4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                g4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < n) {
                    v4[u] = *reinterpret_cast<const float4*>(x + r * C + c4 * 4);
                    if (MODE == 1) g4[u] = *reinterpret_cast<const float4*>(dy + r * C + c4 * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 v = v4[u];
                if (MODE == 0) {
                    // rows past the end contribute exact zeros
                    a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
                    b[0] += (double)v.x * v.x; b[1] += (double)v.y * v.y; b[2] += (double)v.z * v.z; b[3] += (double)v.w * v.w;
                } else {
                    float4 g = g4[u];
                    if (relu) {
                        if (!(v.x * sc.x + sf.x > 0.f)) g.x = 0.f;
                        if (!(v.y * sc.y + sf.y > 0.f)) g.y = 0.f;
                        if (!(v.z * sc.z + sf.z > 0.f)) g.z = 0.f;
                        if (!(v.w * sc.w + sf.w > 0.f)) g.w = 0.f;
                    }
                    a[0] += g.x; a[1] += g.y; a[2] += g.z; a[3] += g.w;
                    b[0] += (double)g.x * ((v.x - mu.x) * is.x); b[1] += (double)g.y * ((v.y - mu.y) * is.y);
                    b[2] += (double)g.z * ((v.z - mu.z) * is.z); b[3] += (double)g.w * ((v.w - mu.w) * is.w);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { sh[tid * 8 + j] = a[j]; sh[tid * 8 + 4 + j] = b[j]; }
    __syncthreads();
    if (tid < lpr) {
        double ta[4] = {0, 0, 0, 0}, tb[4] = {0, 0, 0, 0};
        for (int s = 0; s < rpb; ++s) {
            const int o = (s * lpr + tid) * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) { ta[j] += sh[o + j]; tb[j] += sh[o + 4 + j]; }
        }
        // per-block partial row (1024-way same-address fp64 atomics cost ~100 us on the big levels)
        double* out = sums + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            out[tid * 4 + j] = ta[j];
            out[C + tid * 4 + j] = tb[j];
        }
    }
}

// Statistics of a convolution output from the per-tile partial sums its epilogue wrote (spconv_gmm_k: float [n_tiles][2][C] =
// sum x | sum x^2 over the <= 64 rows of a tile): no pass over x at all.  Workgroup b adds the tiles b, b + G, b + 2G, ... in
// fp64; bn_sum_k finishes.
__global__ __launch_bounds__(256) void bn_partials_k(const float* __restrict__ partial, int64_t n_tiles, int C, double* ws) {
    __shared__ double sh[256];
    const int C2 = 2 * C, tid = threadIdx.x;
    double* out = ws + (int64_t)blockIdx.x * C2;
    for (int c0 = 0; c0 < C2; c0 += 256) {                 // column chunks of up to 256
        const int w = min(256, C2 - c0), rpp = 256 / w;    // rows per pass
        const int col = c0 + tid % w, slot = tid / w;
        double a0 = 0.0, a1 = 0.0;
        if (slot < rpp) {
            const int64_t step = (int64_t)gridDim.x * rpp;
            int64_t r = (int64_t)blockIdx.x * rpp + slot;
            for (; r + step < n_tiles; r += 2 * step) {    // two independent loads in flight
                a0 += (double)partial[r * C2 + col];
                a1 += (double)partial[(r + step) * C2 + col];
            }
            if (r < n_tiles) a0 += (double)partial[r * C2 + col];
        }
        sh[tid] = a0 + a1;
        __syncthreads();
        if (tid < w) {
            double t = 0.0;
            for (int s_ = 0; s_ < rpp; ++s_) t += sh[s_ * w + tid];
            out[c0 + tid] = t;
        }
        __syncthreads();
    }
}


__global__ void bn_finalize_k(const double* __restrict__ sums, double count, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float eps, float momentum, float* running_mean,
                              float* running_var, int C, float* mean, float* invstd, float* scale, float* shift,
                              int64_t* num_batches_tracked) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    if (c >= C) return;
    if (!(count > 0.0)) count = sums[2 * C];      // count travels with the (all-reduced) sums
    const double m = sums[c] / count;
    double var = sums[C + c] / count - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float mf = (float)m;
    mean[c] = mf;
    invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - mf * sc;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mf;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// A bf16 SHADOW of an output ([n][C] bf16, rounded to nearest even, C % 32 == 0) can be written next to the fp32 tensor: the
// sparse convolutions that gather these rows with bf16 MFMA operands (BASELINE configs[2]) round every row to exactly these values
// each time they gather it -- from the shadow they read half the bytes, already in MFMA fragment shape (spconv_wg.hip PR = 3).
// Inside each 32-channel group the shadow is in FRAGMENT ORDER: the 16 bytes at byte 16 q hold channels 4q .. 4q+3 and
// 16 + 4q .. 16 + 4q + 3 -- the eight reduction elements lane group q of v_mfma_f32_16x16x32_bf16 takes in the fp32-row kernels
// (spconv.hip: k = 8q + e <-> channel 16 (e >> 2) + 4q + (e & 3)), so both kinds of kernel multiply the same operands in the same
// k slots, share the packed weights, and return identical bits.
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
__device__ __forceinline__ void store_shadow(__bf16* dst, int64_t row, int c4, int C4, const float4& o) {
    const int g32 = c4 >> 3, half = (c4 >> 2) & 1, q = c4 & 3;          // channels 4 c4 .. 4 c4 + 3 = group g32, 16-channel half, quad q
    reinterpret_cast<bf16x4_t*>(dst)[row * C4 + g32 * 8 + q * 2 + half] = bf16x4_t{(__bf16)o.x, (__bf16)o.y, (__bf16)o.z, (__bf16)o.w};
}

__global__ __launch_bounds__(256) void bn_apply_k(const float* __restrict__ x, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, int relu, int64_t n4, int C4, float* y, __bf16* yb) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 sc = reinterpret_cast<const float4*>(scale)[c4];
        const float4 sf = reinterpret_cast<const float4*>(shift)[c4];
        float4 o = make_float4(v.x * sc.x + sf.x, v.y * sc.y + sf.y, v.z * sc.z + sf.z, v.w * sc.w + sf.w);
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        reinterpret_cast<float4*>(y)[i] = o;
        if (yb) store_shadow(yb, i / C4, c4, C4, o);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                      const double* __restrict__ sums, double inv_count, int64_t n4, int C,
                                                      float* dx, float* dgamma, float* dbeta, const double* __restrict__ count_src,
                                                      const float* __restrict__ addend, __bf16* dxb) {
    const int C4 = C >> 2;
    if (!(inv_count > 0.0)) inv_count = 1.0 / (count_src ? count_src[0] : sums[2 * C]);
    if (blockIdx.x == 0 && dgamma) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            dbeta[c] = (float)sums[c];
            dgamma[c] = (float)sums[C + c];
        }
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        float4 g = reinterpret_cast<const float4*>(dy)[i];
        const float4 mu = reinterpret_cast<const float4*>(mean)[c4];
        const float4 is = reinterpret_cast<const float4*>(invstd)[c4];
        const float4 sc = reinterpret_cast<const float4*>(scale)[c4];
        const float4 sf = reinterpret_cast<const float4*>(shift)[c4];
        if (relu) {
            if (!(v.x * sc.x + sf.x > 0.f)) g.x = 0.f;
            if (!(v.y * sc.y + sf.y > 0.f)) g.y = 0.f;
            if (!(v.z * sc.z + sf.z > 0.f)) g.z = 0.f;
            if (!(v.w * sc.w + sf.w > 0.f)) g.w = 0.f;
        }
        const int c = c4 * 4;
        const float m1x = (float)(sums[c] * inv_count), m1y = (float)(sums[c + 1] * inv_count);
        const float m1z = (float)(sums[c + 2] * inv_count), m1w = (float)(sums[c + 3] * inv_count);
        const float m2x = (float)(sums[C + c] * inv_count), m2y = (float)(sums[C + c + 1] * inv_count);
        const float m2z = (float)(sums[C + c + 2] * inv_count), m2w = (float)(sums[C + c + 3] * inv_count);
        float4 o;
        o.x = sc.x * (g.x - m1x - (v.x - mu.x) * is.x * m2x);
        o.y = sc.y * (g.y - m1y - (v.y - mu.y) * is.y * m2y);
        o.z = sc.z * (g.z - m1z - (v.z - mu.z) * is.z * m2z);
        o.w = sc.w * (g.w - m1w - (v.w - mu.w) * is.w * m2w);
        if (addend) {        // a second gradient of x (the identity branch of a residual block / U-Net skip) joins here instead of in an add kernel
            const float4 e = reinterpret_cast<const float4*>(addend)[i];
            o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
        }
        reinterpret_cast<float4*>(dx)[i] = o;
        if (dxb) store_shadow(dxb, i / C4, c4, C4, o);
    }
}

static bool bn_ok(int64_t n, int C) { return n > 0 && C >= 4 && C <= 256 && C % 4 == 0; }
static bool shadow_ok(const void* sh, int C) { return !sh || C % 32 == 0; }
static unsigned ew_grid(int64_t n4) {
    int64_t g = ceil_div(n4, 256);
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace u3d

using namespace u3d;

extern "C" {

static int bn_grid(int64_t n, int C) {
    const int rpb = 256 / (C / 4);
    int64_t g = ceil_div(n, (int64_t)rpb * 16);
    return (int)(g < 1 ? 1 : (g > 256 ? 256 : g));
}
static int bn_partials_grid(int64_t n_tiles, int C) {
    const int rpp = 2 * C >= 256 ? 1 : 256 / (2 * C);
    int64_t g = ceil_div(n_tiles, (int64_t)rpp * 16);         // ~16 tiles per thread
    return (int)(g < 1 ? 1 : (g > 128 ? 128 : g));
}

int64_t u3d_bn_ws_bytes(int C) { return (int64_t)256 * 2 * C * sizeof(double) + 64; }

static int bn_finish_launch(const void* ws, int nblk, int C, const BnFin& f, hipStream_t s) {
    hipLaunchKernelGGL(bn_sum_k, dim3((C + 3) / 4), dim3(256), 0, s, (const double*)ws, nblk, C, f);
    return check_launch("bn_sum");
}

static BnFin bn_fin(double* sums, double rows, int set_rows) {
    BnFin f;
    f.sums = sums; f.rows = rows; f.set_rows = set_rows; f.gamma = nullptr; f.beta = nullptr; f.eps = 0.f; f.momentum = 0.f;
    f.running_mean = nullptr; f.running_var = nullptr; f.st = nullptr; f.nbt = nullptr;
    return f;
}

// statistics of x [n][C] (pass over x), or -- partial != NULL -- from the per-tile sums a convolution epilogue wrote
static int bn_stats_launch(const float* x, int64_t n, int C, const float* partial, int64_t n_tiles, const BnFin& f, void* ws, hipStream_t s) {
    if (partial) {
        if (n_tiles <= 0) return U3D_EINVAL;
        ProfScope prof(U3D_K_BN, s, (double)n_tiles * C * 8);
        const int g = bn_partials_grid(n_tiles, C);
        hipLaunchKernelGGL(bn_partials_k, dim3(g), dim3(256), 0, s, partial, n_tiles, C, (double*)ws);
        return bn_finish_launch(ws, g, C, f, s);
    }
    ProfScope prof(U3D_K_BN, s, (double)n * C * 4);
    const int g = bn_grid(n, C);
    hipLaunchKernelGGL(bn_reduce_k<0>, dim3(g), dim3(256), 0, s, x, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, n, C, (double*)ws);
    return bn_finish_launch(ws, g, C, f, s);
}

int u3d_bn_stats(const float* x, int64_t n, int C, const float* partial, int64_t n_tiles, double* sums, void* ws, u3d_stream_t stream) {
    if ((!x && !partial) || !sums || !ws || !bn_ok(n, C)) return U3D_EINVAL;
    return bn_stats_launch(x, n, C, partial, n_tiles, bn_fin(sums, (double)n, 1), ws, (hipStream_t)stream);
}

int u3d_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, int C, float* mean, float* invstd, float* scale,
                    float* shift, int64_t* num_batches_tracked, u3d_stream_t stream) {
    if (!sums || !gamma || !beta || !mean || !invstd || !scale || !shift || C <= 0) return U3D_EINVAL;
    hipLaunchKernelGGL(bn_finalize_k, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, sums, count, gamma, beta, eps,
                       momentum, running_mean, running_var, C, mean, invstd, scale, shift, num_batches_tracked);
    return check_launch("bn_finalize");
}

int u3d_bn_apply(const float* x, const float* scale, const float* shift, int relu, int64_t n, int C, float* y, void* y_bf16,
                 u3d_stream_t stream) {
    if (!x || !scale || !shift || !y || !bn_ok(n, C) || !shadow_ok(y_bf16, C)) return U3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_BN, s, (double)n * C * (y_bf16 ? 10 : 8));
    const int64_t n4 = n * (C / 4);
    hipLaunchKernelGGL(bn_apply_k, dim3(ew_grid(n4)), dim3(256), 0, s, x, scale, shift, relu, n4, C / 4, y, (__bf16*)y_bf16);
    return check_launch("bn_apply");
}

int u3d_bn_bwd_stats(const float* x, const float* dy, const float* mean, const float* invstd, const float* scale,
                     const float* shift, int relu, int64_t n, int C, double* sums, void* ws, u3d_stream_t stream) {
    if (!x || !dy || !mean || !invstd || !scale || !shift || !sums || !ws || !bn_ok(n, C)) return U3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_BN, s, (double)n * C * 8);
    const int g = bn_grid(n, C);
    const BnFin f = bn_fin(sums, 0.0, 0);
    hipLaunchKernelGGL(bn_reduce_k<1>, dim3(g), dim3(256), 0, s, x, dy, mean, invstd, scale, shift, relu, n, C, (double*)ws);
    return bn_finish_launch(ws, g, C, f, s);
}

int u3d_bn_bwd_apply(const float* x, const float* dy, const float* mean, const float* invstd, const float* scale,
                     const float* shift, int relu, const double* sums, double count, int64_t n, int C, float* dx, void* dx_bf16,
                     float* dgamma, float* dbeta, const float* addend, u3d_stream_t stream) {
    if (!x || !dy || !mean || !invstd || !scale || !shift || !sums || !dx || !bn_ok(n, C) || !shadow_ok(dx_bf16, C)) return U3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_BN, s, (double)n * C * 12);
    const int64_t n4 = n * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_k, dim3(ew_grid(n4)), dim3(256), 0, s, x, dy, mean, invstd, scale, shift, relu, sums,
                       1.0 / count, n4, C, dx, dgamma, dbeta, (const double*)nullptr, addend, (__bf16*)dx_bf16);
    return check_launch("bn_bwd_apply");
}

int u3d_bn_forward(const float* x, int64_t n, int C, const float* partial, int64_t n_tiles, const float* gamma, const float* beta, float eps,
                   float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, int relu, float* y, void* y_bf16,
                   float* st, double* sums, void* ws, u3d_stream_t stream) {
    // statistics (per-workgroup partial rows) -> bn_sum_k (sums them and finalizes the layer) -> apply
    if (!x || !sums || !ws || !gamma || !beta || !st || !y || !bn_ok(n, C)) return U3D_EINVAL;
    BnFin f = bn_fin(sums, (double)n, 1);
    f.gamma = gamma; f.beta = beta; f.eps = eps; f.momentum = momentum; f.running_mean = running_mean; f.running_var = running_var;
    f.st = st; f.nbt = num_batches_tracked;
    int rc = bn_stats_launch(x, n, C, partial, n_tiles, f, ws, (hipStream_t)stream);
    if (rc) return rc;
    return u3d_bn_apply(x, st + 2 * C, st + 3 * C, relu, n, C, y, y_bf16, stream);
}

int u3d_bn_backward(const float* x, const float* dy, const float* st, int relu, const double* fwd_sums, double* sums, int64_t n, int C,
                    float* dx, void* dx_bf16, float* dgamma, float* dbeta, const float* addend, void* ws, u3d_stream_t stream) {
    // sums[0..2C) are overwritten; sums[2C] (the row count) is taken from the forward pass's vector
    if (!fwd_sums) return U3D_EINVAL;
    int rc = u3d_bn_bwd_stats(x, dy, st, st + C, st + 2 * C, st + 3 * C, relu, n, C, sums, ws, stream);
    if (rc) return rc;
    if (!dx || !shadow_ok(dx_bf16, C)) return U3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;       // the row count is read from the forward pass's vector (no copy launch)
    ProfScope prof(U3D_K_BN, s, (double)n * C * 12);
    const int64_t n4 = n * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_k, dim3(ew_grid(n4)), dim3(256), 0, s, x, dy, (const float*)st, (const float*)(st + C), (const float*)(st + 2 * C),
                       (const float*)(st + 3 * C), relu, (const double*)sums, -1.0, n4, C, dx, dgamma, dbeta, fwd_sums + 2 * C, addend,
                       (__bf16*)dx_bf16);
    return check_launch("bn_backward");
}

}  // extern "C"
