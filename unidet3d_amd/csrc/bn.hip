// K9 BatchNorm(+ReLU) over voxel rows, training mode, on gfx950.  HBM-bound: every pass streams
// [n, C] fp32 with 16-byte coalesced loads; statistics are accumulated in fp64 so that
// var = E[x^2] - mean^2 is as accurate as torch's two-pass kernel, and so that the per-rank partial
// sums can be all-reduced across data-parallel ranks (SyncBatchNorm) between stats and apply.
// Replaces nn.SyncBatchNorm / nn.BatchNorm1d(eps=1e-4, momentum=0.1) + nn.ReLU at
// unidet3d/spconv_unet.py:42,49,119-124,147,177 and unidet3d/unidet3d.py:104-111.
#include "u3d_common.h"

namespace u3d {

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

// sums[c] = sum_b partial[b][c] in a fixed order (deterministic); optionally sums[2C] = rows.
// One wave per column: lane l adds rows l, l+64, ... then a fixed shuffle tree.
__global__ __launch_bounds__(256) void bn_sum_partials_k(const double* __restrict__ partial, int nblk, int C2, double rows,
                                                         int set_rows, double* __restrict__ sums) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c < C2) {
        double t = 0.0;
        for (int b = lane; b < nblk; b += 64) t += partial[(int64_t)b * C2 + c];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
        if (lane == 0) sums[c] = t;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && set_rows) sums[C2] = rows;
}

// bn_sum_partials_k for the two statistics of channel c + bn_finalize_k of that channel in one launch (one wave per channel)
__global__ __launch_bounds__(256) void bn_sum_finalize_k(const double* __restrict__ partial, int nblk, int C, double rows, double* __restrict__ sums,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                                         float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                                                         float* shift, int64_t* num_batches_tracked) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sums[2 * C] = rows;
        if (num_batches_tracked) *num_batches_tracked += 1;
    }
    if (c >= C) return;
    double t1 = 0.0, t2 = 0.0;
    for (int b = lane; b < nblk; b += 64) {
        t1 += partial[(int64_t)b * 2 * C + c];
        t2 += partial[(int64_t)b * 2 * C + C + c];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { t1 += __shfl_xor(t1, d, 64); t2 += __shfl_xor(t2, d, 64); }
    if (lane) return;
    sums[c] = t1;
    sums[C + c] = t2;
    const double m = t1 / rows;
    double var = t2 / rows - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float mf = (float)m;
    mean[c] = mf;
    invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - mf * sc;
    if (running_mean) {
        const double unbiased = rows > 1.0 ? var * rows / (rows - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mf;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
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

__global__ __launch_bounds__(256) void bn_apply_k(const float* __restrict__ x, const float* __restrict__ scale,
                                                  const float* __restrict__ shift, int relu, int64_t n4, int C4, float* y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 sc = reinterpret_cast<const float4*>(scale)[c4];
        const float4 sf = reinterpret_cast<const float4*>(shift)[c4];
        float4 o = make_float4(v.x * sc.x + sf.x, v.y * sc.y + sf.y, v.z * sc.z + sf.z, v.w * sc.w + sf.w);
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        reinterpret_cast<float4*>(y)[i] = o;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                      const double* __restrict__ sums, double inv_count, int64_t n4, int C,
                                                      float* dx, float* dgamma, float* dbeta, const double* __restrict__ count_src,
                                                      const float* __restrict__ addend) {
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
    }
}

static bool bn_ok(int64_t n, int C) { return n > 0 && C >= 4 && C <= 256 && C % 4 == 0; }
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

int64_t u3d_bn_ws_bytes(int C) { return (int64_t)256 * 2 * C * sizeof(double) + 64; }

int u3d_bn_stats(const float* x, int64_t n, int C, double* sums, void* ws, u3d_stream_t stream) {
    if (!x || !sums || !ws || !bn_ok(n, C)) return U3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_BN, s, (double)n * C * 4);
    const int g = bn_grid(n, C);
    hipLaunchKernelGGL(bn_reduce_k<0>, dim3(g), dim3(256), 0, s, x, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, n, C, (double*)ws);
    hipLaunchKernelGGL(bn_sum_partials_k, dim3((2 * C + 3) / 4), dim3(256), 0, s, (const double*)ws, g, 2 * C, (double)n, 1, sums);
    return check_launch("bn_stats");
}

int u3d_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, int C, float* mean, float* invstd, float* scale,
                    float* shift, int64_t* num_batches_tracked, u3d_stream_t stream) {
    if (!sums || !gamma || !beta || !mean || !invstd || !scale || !shift || C <= 0) return U3D_EINVAL;
    hipLaunchKernelGGL(bn_finalize_k, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, sums, count, gamma, beta, eps,
                       momentum, running_mean, running_var, C, mean, invstd, scale, shift, num_batches_tracked);
    return check_launch("bn_finalize");
}

int u3d_bn_apply(const float* x, const float* scale, const float* shift, int relu, int64_t n, int C, float* y,
                 u3d_stream_t stream) {
    if (!x || !scale || !shift || !y || !bn_ok(n, C)) return U3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_BN, s, (double)n * C * 8);
    const int64_t n4 = n * (C / 4);
    hipLaunchKernelGGL(bn_apply_k, dim3(ew_grid(n4)), dim3(256), 0, s, x, scale, shift, relu, n4, C / 4, y);
    return check_launch("bn_apply");
}

int u3d_bn_bwd_stats(const float* x, const float* dy, const float* mean, const float* invstd, const float* scale,
                     const float* shift, int relu, int64_t n, int C, double* sums, void* ws, u3d_stream_t stream) {
    if (!x || !dy || !mean || !invstd || !scale || !shift || !sums || !ws || !bn_ok(n, C)) return U3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_BN, s, (double)n * C * 8);
    const int g = bn_grid(n, C);
    hipLaunchKernelGGL(bn_reduce_k<1>, dim3(g), dim3(256), 0, s, x, dy, mean, invstd, scale, shift, relu, n, C, (double*)ws);
    hipLaunchKernelGGL(bn_sum_partials_k, dim3((2 * C + 3) / 4), dim3(256), 0, s, (const double*)ws, g, 2 * C, 0.0, 0, sums);
    return check_launch("bn_bwd_stats");
}

int u3d_bn_bwd_apply(const float* x, const float* dy, const float* mean, const float* invstd, const float* scale,
                     const float* shift, int relu, const double* sums, double count, int64_t n, int C, float* dx,
                     float* dgamma, float* dbeta, const float* addend, u3d_stream_t stream) {
    if (!x || !dy || !mean || !invstd || !scale || !shift || !sums || !dx || !bn_ok(n, C)) return U3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_BN, s, (double)n * C * 12);
    const int64_t n4 = n * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_k, dim3(ew_grid(n4)), dim3(256), 0, s, x, dy, mean, invstd, scale, shift, relu, sums,
                       1.0 / count, n4, C, dx, dgamma, dbeta, (const double*)nullptr, addend);
    return check_launch("bn_bwd_apply");
}

int u3d_bn_forward(const float* x, int64_t n, int C, const float* gamma, const float* beta, float eps, float momentum,
                   float* running_mean, float* running_var, int64_t* num_batches_tracked, int relu, float* y, float* st, double* sums,
                   void* ws, u3d_stream_t stream) {
    // statistics (per-block partials) -> one launch that sums the partials of a channel AND finalises it -> apply
    if (!x || !sums || !ws || !gamma || !beta || !st || !y || !bn_ok(n, C)) return U3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope prof(U3D_K_BN, s, (double)n * C * 4);
        const int g = bn_grid(n, C);
        hipLaunchKernelGGL(bn_reduce_k<0>, dim3(g), dim3(256), 0, s, x, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, n, C, (double*)ws);
        hipLaunchKernelGGL(bn_sum_finalize_k, dim3((C + 3) / 4), dim3(256), 0, s, (const double*)ws, g, C, (double)n, sums, gamma, beta, eps,
                           momentum, running_mean, running_var, st, st + C, st + 2 * C, st + 3 * C, num_batches_tracked);
        int rc = check_launch("bn_forward");
        if (rc) return rc;
    }
    return u3d_bn_apply(x, st + 2 * C, st + 3 * C, relu, n, C, y, stream);
}

int u3d_bn_backward(const float* x, const float* dy, const float* st, int relu, const double* fwd_sums, double* sums, int64_t n, int C,
                    float* dx, float* dgamma, float* dbeta, const float* addend, void* ws, u3d_stream_t stream) {
    // sums[0..2C) are overwritten; sums[2C] (the row count) is taken from the forward pass's vector
    if (!fwd_sums) return U3D_EINVAL;
    int rc = u3d_bn_bwd_stats(x, dy, st, st + C, st + 2 * C, st + 3 * C, relu, n, C, sums, ws, stream);
    if (rc) return rc;
    if (!dx) return U3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;       // the row count is read from the forward pass's vector (no copy launch)
    ProfScope prof(U3D_K_BN, s, (double)n * C * 12);
    const int64_t n4 = n * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_k, dim3(ew_grid(n4)), dim3(256), 0, s, x, dy, (const float*)st, (const float*)(st + C), (const float*)(st + 2 * C),
                       (const float*)(st + 3 * C), relu, (const double*)sums, -1.0, n4, C, dx, dgamma, dbeta, fwd_sums + 2 * C, addend);
    return check_launch("bn_backward");
}

}  // extern "C"
