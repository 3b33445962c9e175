// K15: LayerNorm of the decoder (unidet3d/encoder.py:21,38-40,61,78-79,140,167: nn.LayerNorm(d_model) after every
// residual add and in front of the prediction heads), forward with the residual add fused in, and backward.
// HBM-bound: forward reads x (+res) and writes y (+ the sum kept for backward); backward reads the sum and dy, writes
// dx; the parameter gradients are per-workgroup partials summed in a fixed order (deterministic, no atomics).
// One wave per row; a lane owns the float4 columns lane, lane+64, ... (C % 4 == 0, C <= 1024).
#include <stdlib.h>

#include "u3d_common.h"

namespace u3d {

constexpr int LN_MAXV = 4;       // float4 per lane: C <= 1024

typedef __attribute__((ext_vector_type(2))) __bf16 ln_bf16x2;
__device__ __forceinline__ unsigned ln_pack_bf16(float lo, float hi) { return __builtin_bit_cast(unsigned, ln_bf16x2{(__bf16)lo, (__bf16)hi}); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int NV>
__global__ __launch_bounds__(256) void layer_norm_fwd_k(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int64_t M, int C, float eps, float* __restrict__ sum_out,
                                                        float* __restrict__ y, float* __restrict__ stats, uint2* __restrict__ y16) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int c4 = C >> 2;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = lane + 64 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < c4) {
            v[i] = reinterpret_cast<const float4*>(x + row * C)[j];
            if (res) {
                const float4 r = reinterpret_cast<const float4*>(res + row * C)[j];
                v[i].x += r.x; v[i].y += r.y; v[i].z += r.z; v[i].w += r.w;
                reinterpret_cast<float4*>(sum_out + row * C)[j] = v[i];
            }
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (lane + 64 * i < c4) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = lane + 64 * i;
        if (j < c4) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[j], b = reinterpret_cast<const float4*>(beta)[j];
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x; o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z; o.w = (v[i].w - mean) * rstd * g.w + b.w;
            reinterpret_cast<float4*>(y + row * C)[j] = o;
            if (y16) y16[row * c4 + j] = make_uint2(ln_pack_bf16(o.x, o.y), ln_pack_bf16(o.z, o.w));      // the copy the next GEMM streams (K14b)
        }
    }
    if (lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
}

// dx = rstd * (g - mean(g) - xh * mean(g * xh)),  g = dy * gamma,  xh = (s - mean) * rstd
// partial[block][0][C] = sum_rows dy * xh (dgamma),  partial[block][1][C] = sum_rows dy (dbeta)
template <int NV>
__global__ __launch_bounds__(256) void layer_norm_bwd_k(const float* __restrict__ s, const float* __restrict__ dy, const float* __restrict__ gamma,
                                                        const float* __restrict__ stats, int64_t M, int C, float* __restrict__ dx,
                                                        float* __restrict__ partial, uint2* __restrict__ dx16,
                                                        const float* __restrict__ dy2, const float* __restrict__ dy3) {
    __shared__ float red[4 * 2 * LN_MAXV * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c4 = C >> 2;
    float4 gm[NV], dg[NV], db[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = lane + 64 * i;
        gm[i] = j < c4 ? reinterpret_cast<const float4*>(gamma)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[i] = db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
        const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
        float4 xh[NV], g[NV];
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int j = lane + 64 * i;
            xh[i] = g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < c4) {
                const float4 a = reinterpret_cast<const float4*>(s + row * C)[j];
                float4 d = reinterpret_cast<const float4*>(dy + row * C)[j];
                if (dy2) {            // the result had more than one consumer: their gradients are summed here, in a fixed order
                    const float4 e = reinterpret_cast<const float4*>(dy2 + row * C)[j];
                    d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w;
                }
                if (dy3) {
                    const float4 e = reinterpret_cast<const float4*>(dy3 + row * C)[j];
                    d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w;
                }
                xh[i] = make_float4((a.x - mean) * rstd, (a.y - mean) * rstd, (a.z - mean) * rstd, (a.w - mean) * rstd);
                g[i] = make_float4(d.x * gm[i].x, d.y * gm[i].y, d.z * gm[i].z, d.w * gm[i].w);
                dg[i].x += d.x * xh[i].x; dg[i].y += d.y * xh[i].y; dg[i].z += d.z * xh[i].z; dg[i].w += d.w * xh[i].w;
                db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
                sg += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                sgx += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
            }
        }
        const float mg = wave_sum(sg) / (float)C, mgx = wave_sum(sgx) / (float)C;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int j = lane + 64 * i;
            if (j < c4) {
                const float4 o = make_float4(rstd * (g[i].x - mg - xh[i].x * mgx), rstd * (g[i].y - mg - xh[i].y * mgx),
                                             rstd * (g[i].z - mg - xh[i].z * mgx), rstd * (g[i].w - mg - xh[i].w * mgx));
                reinterpret_cast<float4*>(dx + row * C)[j] = o;
                if (dx16) dx16[row * c4 + j] = make_uint2(ln_pack_bf16(o.x, o.y), ln_pack_bf16(o.z, o.w));
            }
        }
    }
    // 4 waves -> one partial per workgroup, fixed order
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        reinterpret_cast<float4*>(red)[(wave * 2 + 0) * (LN_MAXV * 64) + i * 64 + lane] = dg[i];
        reinterpret_cast<float4*>(red)[(wave * 2 + 1) * (LN_MAXV * 64) + i * 64 + lane] = db[i];
    }
    __syncthreads();
    if (wave < 2) {                      // wave 0: dgamma, wave 1: dbeta
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int j = lane + 64 * i;
            if (j < c4) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float4 u = reinterpret_cast<const float4*>(red)[(w * 2 + wave) * (LN_MAXV * 64) + i * 64 + lane];
                    t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
                }
                reinterpret_cast<float4*>(partial + ((int64_t)blockIdx.x * 2 + wave) * C)[j] = t;
            }
        }
    }
}

// dgamma / dbeta = sum of the per-workgroup partials, fixed order.  One workgroup of 16 waves per 64 columns of the [2C] output:
// wave w adds partial rows w, w+16, ... (256-byte coalesced reads, four independent accumulators), then the 16 wave sums are
// combined in LDS in a fixed order.  (The single-thread-per-column form walked all 512 rows serially: 33 us per call.)
__global__ __launch_bounds__(1024) void layer_norm_reduce_k(const float* __restrict__ partial, int nblocks, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;                        // over 2*C: [0, C) = dgamma, [C, 2C) = dbeta (C % 64 == 0 not required)
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (j < 2 * C) {
        const int which = j / C, c = j % C;
        int b = wave;
        for (; b + 48 < nblocks; b += 64) {
            v0 += partial[((int64_t)(b + 0) * 2 + which) * C + c];
            v1 += partial[((int64_t)(b + 16) * 2 + which) * C + c];
            v2 += partial[((int64_t)(b + 32) * 2 + which) * C + c];
            v3 += partial[((int64_t)(b + 48) * 2 + which) * C + c];
        }
        for (; b < nblocks; b += 16) v0 += partial[((int64_t)b * 2 + which) * C + c];
    }
    red[wave][lane] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (wave == 0 && j < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w][lane];
        (j / C ? dbeta : dgamma)[j % C] = t;
    }
}

// workgroups of the backward pass (each walks rows with a stride and leaves one partial dgamma / dbeta row): 512, or 1024 for the head's
// tall input (tools/prof_ln.py, backward + reduce, us at 256 | 512 | 1024 | 2048 workgroups: M = 16 937: 23.3 | 17.9 | 18.0 | 24.2;
// 24 600: 30.8 | 21.8 | 21.7 | 27.9; 118 559: 158.6 | 99.8 | 75.3 | 80.6)
static int ln_blocks(int64_t M) {
    static const int forced = [] { const char* e = getenv("U3D_LN_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
    const int cap = forced ? forced : (M >= 65536 ? 1024 : 512);
    const int64_t b = ceil_div(M, 4);
    return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace u3d

using namespace u3d;

extern "C" {

int64_t u3d_layer_norm_ws_bytes(int64_t M, int C) { return (int64_t)ln_blocks(M) * 2 * C * 4 + 256; }

int u3d_layer_norm_fwd(const float* x, const float* res, const float* gamma, const float* beta, int64_t M, int C, float eps,
                       float* sum_out, float* y, float* stats, u3d_stream_t stream) {
    return u3d_layer_norm_fwd_b16(x, res, gamma, beta, M, C, eps, sum_out, y, nullptr, stats, stream);
}

int u3d_layer_norm_fwd_b16(const float* x, const float* res, const float* gamma, const float* beta, int64_t M, int C, float eps,
                           float* sum_out, float* y, void* y16, float* stats, u3d_stream_t stream) {
    if (!x || !gamma || !beta || !y || !stats || M < 0 || C <= 0 || (res && !sum_out)) return U3D_EINVAL;
    if (C % 4 || C > 256 * LN_MAXV) { set_error("layer_norm: C=%d unsupported (multiple of 4, <= %d)", C, 256 * LN_MAXV); return U3D_EUNSUPPORTED; }
    if (M == 0) return U3D_OK;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)ceil_div(M, 4));
    const int nv = (int)ceil_div(C / 4, 64);
#define U3D_LN(NV) hipLaunchKernelGGL(layer_norm_fwd_k<NV>, grid, dim3(256), 0, s, x, res, gamma, beta, M, C, eps, sum_out, y, stats, (uint2*)y16)
    if (nv == 1) U3D_LN(1); else if (nv == 2) U3D_LN(2); else U3D_LN(4);
#undef U3D_LN
    return check_launch("layer_norm_fwd");
}

int u3d_layer_norm_bwd(const float* s_in, const float* dy, const float* gamma, const float* stats, int64_t M, int C, float* dx,
                       float* dgamma, float* dbeta, void* ws, u3d_stream_t stream) {
    return u3d_layer_norm_bwd_b16(s_in, dy, gamma, stats, M, C, dx, nullptr, dgamma, dbeta, ws, stream);
}

int u3d_layer_norm_bwd_b16(const float* s_in, const float* dy, const float* gamma, const float* stats, int64_t M, int C, float* dx, void* dx16,
                           float* dgamma, float* dbeta, void* ws, u3d_stream_t stream) {
    return u3d_layer_norm_bwd_sum(s_in, dy, nullptr, nullptr, gamma, stats, M, C, dx, dx16, dgamma, dbeta, ws, stream);
}

int u3d_layer_norm_bwd_sum(const float* s_in, const float* dy, const float* dy2, const float* dy3, const float* gamma, const float* stats,
                           int64_t M, int C, float* dx, void* dx16, float* dgamma, float* dbeta, void* ws, u3d_stream_t stream) {
    if (!s_in || !dy || !gamma || !stats || !dx || !dgamma || !dbeta || !ws || M < 0 || C <= 0) return U3D_EINVAL;
    if (C % 4 || C > 256 * LN_MAXV) { set_error("layer_norm: C=%d unsupported (multiple of 4, <= %d)", C, 256 * LN_MAXV); return U3D_EUNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    const int nb = ln_blocks(M);
    const int nv = (int)ceil_div(C / 4, 64);
#define U3D_LN(NV) hipLaunchKernelGGL(layer_norm_bwd_k<NV>, dim3(nb), dim3(256), 0, s, s_in, dy, gamma, stats, M, C, dx, (float*)ws, (uint2*)dx16, dy2, dy3)
    if (nv == 1) U3D_LN(1); else if (nv == 2) U3D_LN(2); else U3D_LN(4);
#undef U3D_LN
    hipLaunchKernelGGL(layer_norm_reduce_k, dim3((unsigned)ceil_div(2 * C, 64)), dim3(1024), 0, s, (const float*)ws, nb, C, dgamma, dbeta);
    return check_launch("layer_norm_bwd");
}

}  // extern "C"
