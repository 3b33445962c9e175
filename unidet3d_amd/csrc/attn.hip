// K13 self-attention core on gfx950: varlen (cu_seqlens-packed) flash attention, fp32 in / fp32
// accumulate on v_mfma_f32_16x16x4_f32, head_dim 32.  Replaces the math path of
// nn.MultiheadAttention(256, 8, batch_first=True) at unidet3d/encoder.py:19-20,36-37 (called per
// scene in a Python loop by the reference); the n x n score matrix never leaves the CU.
//
// Layout trick used throughout: the C/D fragment of a 16x16 MFMA (col = lane&15, rows = 4*(lane>>4)+r)
// is exactly the A fragment of the next MFMA whose reduction index is D's row index.  So
//   forward:  S^T = K Q^T  (A = K tile from LDS, B = own Q rows in registers)
//             -> softmax over keys = in-lane + two shuffles -> P is already the A operand of O += P V.
//   dQ:       S^T, dP^T = V dO^T (same orientation) -> dS is the A operand of dQ += dS K.
//   dK/dV:    S = Q K^T, dP = dO V^T (A = Q / dO tile from LDS, B = own K / V rows in registers)
//             -> P^T, dS^T are the A operands of dV += P^T dO and dK += dS^T Q.
#include "u3d_common.h"

namespace u3d {

constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
#define U3D_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int ATT_LD = 36;   // 32 + 4 pad floats per staged row

// stage 64 rows x 32 floats of (qkv + col_off) into dst[64][ATT_LD]; rows >= len are zero; optional scale
__device__ __forceinline__ void stage_tile(const float* __restrict__ base, int ld, int row0, int len, float scale, float* dst, int tid) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = tid + j * 256;
        const int r = idx >> 3, c4 = idx & 7;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < len) v = *reinterpret_cast<const float4*>(base + (int64_t)(row0 + r) * ld + c4 * 4);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        *reinterpret_cast<float4*>(dst + r * ATT_LD + c4 * 4) = v;
    }
}

// 1-D launch decode: workgroup b runs on XCD b % 8 (private L2).  XCD x takes the (scene, head) pairs x, x+8, ...
// and all their 64-row tiles back to back, so the K/V (or Q/dO) rows of one (scene, head) stay in that XCD's
// L2 while its tiles stream them (PMC before: 321 MB fetched per forward launch for 49 MB of qkv).
struct AttnWork { int b, h, tile; };
__device__ __forceinline__ AttnWork attn_decode(int H, int B, int n_tiles) {
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int hb = (j / n_tiles) * 8 + x;
    AttnWork w;
    w.tile = j % n_tiles;
    w.h = hb % H;
    w.b = hb / H;          // >= B for the padding workgroups of the last group
    return w;
}
static inline unsigned attn_grid(int H, int B, int n_tiles) { return (unsigned)(((H * B + 7) / 8) * 8 * n_tiles); }

__global__ __launch_bounds__(256) void attn_fwd_k(const float* __restrict__ qkv, const int32_t* __restrict__ cu, int H, float scale,
                                                  float* __restrict__ out, float* __restrict__ lse, int64_t n_total, int B, int n_tiles) {
    __shared__ __attribute__((aligned(16))) float Ks[64 * ATT_LD];
    __shared__ __attribute__((aligned(16))) float Vs[64 * ATT_LD];
    const AttnWork wk_ = attn_decode(H, B, n_tiles);
    const int b = wk_.b, h = wk_.h;
    if (b >= B) return;
    const int start = cu[b], len = cu[b + 1] - start;
    const int q0 = wk_.tile * 64;
    if (q0 >= len) return;
    const int D = H * 32, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, qd = lane >> 4;
    const float* base = qkv + (int64_t)start * ld + h * 32;
    const int qrow = q0 + wave * 16 + i16;
    float4 qf[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        qf[d] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qrow < len) qf[d] = *reinterpret_cast<const float4*>(base + (int64_t)qrow * ld + d * 16 + qd * 4);
        // scores in log2 units: q carries scale * log2(e), so the softmax needs v_exp_f32 only (VALU instructions add to the
        // MFMA time on this hardware -- the loop is written for VALU count)
        qf[d].x *= scale * LOG2E; qf[d].y *= scale * LOG2E; qf[d].z *= scale * LOG2E; qf[d].w *= scale * LOG2E;
    }
    float m = -INFINITY, l = 0.f;
    f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int ntiles = (len + 63) >> 6;
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
        stage_tile(base + D, ld, kt * 64, len, 1.f, Ks, tid);
        stage_tile(base + 2 * D, ld, kt * 64, len, 1.f, Vs, tid);
        __syncthreads();
        f32x4 st[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            st[kb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const float4 a = *reinterpret_cast<const float4*>(Ks + (kb * 16 + i16) * ATT_LD + d * 16 + qd * 4);
                st[kb] = U3D_MFMA(a.x, qf[d].x, st[kb]);
                st[kb] = U3D_MFMA(a.y, qf[d].y, st[kb]);
                st[kb] = U3D_MFMA(a.z, qf[d].z, st[kb]);
                st[kb] = U3D_MFMA(a.w, qf[d].w, st[kb]);
            }
        }
        if (kt == ntiles - 1 && (len & 63)) {          // only the last tile can hold keys past the end (wave-uniform)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * 64 + kb * 16 + qd * 4 + r >= len) st[kb][r] = -INFINITY;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(st[kb][r] - m_new);
                st[kb][r] = p;
                ps += p;
            }
        ps += __shfl_xor(ps, 16, 64);
        ps += __shfl_xor(ps, 32, 64);
        l = l * alpha + ps;
        m = m_new;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ar = __shfl(alpha, qd * 4 + r, 64);
            o[0][r] *= ar;
            o[1][r] *= ar;
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float* vrow = Vs + (kb * 16 + qd * 4 + s) * ATT_LD + i16;
                o[0] = U3D_MFMA(st[kb][s], vrow[0], o[0]);
                o[1] = U3D_MFMA(st[kb][s], vrow[16], o[1]);
            }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float lr = __shfl(l, qd * 4 + r, 64);
        const int row = q0 + wave * 16 + qd * 4 + r;
        if (row < len) {
            const float inv = 1.f / lr;
            float* op = out + (int64_t)(start + row) * D + h * 32 + i16;
            op[0] = o[0][r] * inv;
            op[16] = o[1][r] * inv;
        }
    }
    if (qd == 0 && qrow < len) lse[(int64_t)h * n_total + start + qrow] = m * LN2 + __logf(l);      // natural-log units
}

// delta[h][i] = sum_d dO[i][h*32+d] * O[i][h*32+d]
__global__ __launch_bounds__(256) void attn_delta_k(const float* __restrict__ o, const float* __restrict__ dout, int64_t n, int H, float* delta) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * H) return;
    const int64_t i = idx / H;
    const int h = (int)(idx % H);
    const float4* a = reinterpret_cast<const float4*>(o + i * H * 32 + h * 32);
    const float4* b = reinterpret_cast<const float4*>(dout + i * H * 32 + h * 32);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 x = a[j], y = b[j];
        s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    delta[(int64_t)h * n + i] = s;
}

__global__ __launch_bounds__(256) void attn_bwd_dq_k(const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ lse,
                                                     const float* __restrict__ delta, const int32_t* __restrict__ cu, int H, float scale,
                                                     float* __restrict__ dqkv, int64_t n_total, int B, int n_tiles) {
    __shared__ __attribute__((aligned(16))) float Ks[64 * ATT_LD];
    __shared__ __attribute__((aligned(16))) float Vs[64 * ATT_LD];
    const AttnWork wk_ = attn_decode(H, B, n_tiles);
    const int b = wk_.b, h = wk_.h;
    if (b >= B) return;
    const int start = cu[b], len = cu[b + 1] - start;
    const int q0 = wk_.tile * 64;
    if (q0 >= len) return;
    const int D = H * 32, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, qd = lane >> 4;
    const float* base = qkv + (int64_t)start * ld + h * 32;
    const int qrow = q0 + wave * 16 + i16;
    const bool qok = qrow < len;
    float4 qf[2], dof[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        qf[d] = dof[d] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qok) {
            qf[d] = *reinterpret_cast<const float4*>(base + (int64_t)qrow * ld + d * 16 + qd * 4);
            dof[d] = *reinterpret_cast<const float4*>(dout + (int64_t)(start + qrow) * D + h * 32 + d * 16 + qd * 4);
        }
        qf[d].x *= scale * LOG2E; qf[d].y *= scale * LOG2E; qf[d].z *= scale * LOG2E; qf[d].w *= scale * LOG2E;
    }
    // log2 units; rows past the end get +inf so that exp2(s - lse) = 0 masks them without a select per element
    const float lse_q = qok ? lse[(int64_t)h * n_total + start + qrow] * LOG2E : INFINITY;
    const float del_q = qok ? delta[(int64_t)h * n_total + start + qrow] : 0.f;
    f32x4 dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int ntiles = (len + 63) >> 6;
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
        stage_tile(base + D, ld, kt * 64, len, 1.f, Ks, tid);
        stage_tile(base + 2 * D, ld, kt * 64, len, 1.f, Vs, tid);
        __syncthreads();
        const bool last = kt == ntiles - 1 && (len & 63);          // wave-uniform
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, dp4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const float4 ak = *reinterpret_cast<const float4*>(Ks + (kb * 16 + i16) * ATT_LD + d * 16 + qd * 4);
                const float4 av = *reinterpret_cast<const float4*>(Vs + (kb * 16 + i16) * ATT_LD + d * 16 + qd * 4);
                s4 = U3D_MFMA(ak.x, qf[d].x, s4);   dp4 = U3D_MFMA(av.x, dof[d].x, dp4);
                s4 = U3D_MFMA(ak.y, qf[d].y, s4);   dp4 = U3D_MFMA(av.y, dof[d].y, dp4);
                s4 = U3D_MFMA(ak.z, qf[d].z, s4);   dp4 = U3D_MFMA(av.z, dof[d].z, dp4);
                s4 = U3D_MFMA(ak.w, qf[d].w, s4);   dp4 = U3D_MFMA(av.w, dof[d].w, dp4);
            }
            float ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __builtin_amdgcn_exp2f(s4[r] - lse_q);
                if (last && kt * 64 + kb * 16 + qd * 4 + r >= len) p = 0.f;        // zero-padded keys of the last tile
                ds[r] = p * (dp4[r] - del_q);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float* krow = Ks + (kb * 16 + qd * 4 + s) * ATT_LD + i16;
                dq[0] = U3D_MFMA(ds[s], krow[0], dq[0]);
                dq[1] = U3D_MFMA(ds[s], krow[16], dq[1]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = q0 + wave * 16 + qd * 4 + r;
        if (row < len) {
            float* op = dqkv + (int64_t)(start + row) * ld + h * 32 + i16;
            op[0] = dq[0][r] * scale;
            op[16] = dq[1][r] * scale;
        }
    }
}

__global__ __launch_bounds__(256) void attn_bwd_dkv_k(const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ lse,
                                                      const float* __restrict__ delta, const int32_t* __restrict__ cu, int H, float scale,
                                                      float* __restrict__ dqkv, int64_t n_total, int B, int n_tiles) {
    __shared__ __attribute__((aligned(16))) float Qs[64 * ATT_LD];
    __shared__ __attribute__((aligned(16))) float Os[64 * ATT_LD];
    __shared__ float lse_s[64], del_s[64];
    const AttnWork wk_ = attn_decode(H, B, n_tiles);
    const int b = wk_.b, h = wk_.h;
    if (b >= B) return;
    const int start = cu[b], len = cu[b + 1] - start;
    const int k0 = wk_.tile * 64;
    if (k0 >= len) return;
    const int D = H * 32, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, qd = lane >> 4;
    const float* base = qkv + (int64_t)start * ld + h * 32;
    const float* dobase = dout + (int64_t)start * D + h * 32;
    const int krow = k0 + wave * 16 + i16;
    float4 kf[2], vf[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        kf[d] = vf[d] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (krow < len) {
            kf[d] = *reinterpret_cast<const float4*>(base + (int64_t)krow * ld + D + d * 16 + qd * 4);
            vf[d] = *reinterpret_cast<const float4*>(base + (int64_t)krow * ld + 2 * D + d * 16 + qd * 4);
        }
    }
    f32x4 dk[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int ntiles = (len + 63) >> 6;
    for (int qt = 0; qt < ntiles; ++qt) {
        __syncthreads();
        stage_tile(base, ld, qt * 64, len, scale * LOG2E, Qs, tid);        // log2 units; dK is rescaled by ln 2 at the end
        stage_tile(dobase, D, qt * 64, len, 1.f, Os, tid);
        if (tid < 64) {
            const int q = qt * 64 + tid;
            lse_s[tid] = q < len ? lse[(int64_t)h * n_total + start + q] * LOG2E : INFINITY;   // exp2(s - inf) = 0 masks the row
            del_s[tid] = q < len ? delta[(int64_t)h * n_total + start + q] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, dp4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const float4 aq = *reinterpret_cast<const float4*>(Qs + (qb * 16 + i16) * ATT_LD + d * 16 + qd * 4);
                const float4 ao = *reinterpret_cast<const float4*>(Os + (qb * 16 + i16) * ATT_LD + d * 16 + qd * 4);
                s4 = U3D_MFMA(aq.x, kf[d].x, s4);   dp4 = U3D_MFMA(ao.x, vf[d].x, dp4);
                s4 = U3D_MFMA(aq.y, kf[d].y, s4);   dp4 = U3D_MFMA(ao.y, vf[d].y, dp4);
                s4 = U3D_MFMA(aq.z, kf[d].z, s4);   dp4 = U3D_MFMA(ao.z, vf[d].z, dp4);
                s4 = U3D_MFMA(aq.w, kf[d].w, s4);   dp4 = U3D_MFMA(ao.w, vf[d].w, dp4);
            }
            float p[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = qb * 16 + qd * 4 + r;
                p[r] = __builtin_amdgcn_exp2f(s4[r] - lse_s[qq]);
                ds[r] = p[r] * (dp4[r] - del_s[qq]);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float* orow = Os + (qb * 16 + qd * 4 + s) * ATT_LD + i16;
                const float* qrow = Qs + (qb * 16 + qd * 4 + s) * ATT_LD + i16;
                dv[0] = U3D_MFMA(p[s], orow[0], dv[0]);
                dv[1] = U3D_MFMA(p[s], orow[16], dv[1]);
                dk[0] = U3D_MFMA(ds[s], qrow[0], dk[0]);
                dk[1] = U3D_MFMA(ds[s], qrow[16], dk[1]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = k0 + wave * 16 + qd * 4 + r;
        if (row < len) {
            float* op = dqkv + (int64_t)(start + row) * ld + h * 32 + i16;
            op[D] = dk[0][r] * LN2;
            op[D + 16] = dk[1][r] * LN2;
            op[2 * D] = dv[0][r];
            op[2 * D + 16] = dv[1][r];
        }
    }
}

void attn_fwd_x3_launch(const float* qkv, const int32_t* cu, int B, int max_len, int64_t n_total, int H, float scale, float* out, float* lse,
                        hipStream_t s, int planes);
void attn_bwd_x3_launch(const float* qkv, const float* out, const float* dout, const float* lse, const int32_t* cu, int B, int max_len,
                        int64_t n_total, int H, float scale, float* dqkv, float* delta_ws, hipStream_t s, int planes);

}  // namespace u3d

using namespace u3d;

extern "C" {

int u3d_attn_varlen_fwd(const float* qkv, const int32_t* cu_seqlens, int B, int max_len, int64_t n_total, int H, int hd,
                        float scale, float* out, float* lse, double flops_hint, u3d_stream_t stream) {
    if (!qkv || !cu_seqlens || !out || !lse || B <= 0 || H <= 0 || n_total <= 0) return U3D_EINVAL;
    if (hd != 32) { set_error("attn: head_dim %d unsupported (32 only)", hd); return U3D_EUNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_ATTN_FWD, s, flops_hint);
    if (max_len <= 0) return U3D_OK;
    if (fp32_x3()) {                       // default: fp32 products from three bf16 planes (attn_x3.hip)
        attn_fwd_x3_launch(qkv, cu_seqlens, B, max_len, n_total, H, scale, out, lse, s, 3);
        return check_launch("attn_fwd_x3");
    }
    const int n_tiles = (max_len + 63) / 64;
    hipLaunchKernelGGL(attn_fwd_k, dim3(attn_grid(H, B, n_tiles)), dim3(256), 0, s, qkv, cu_seqlens, H, scale, out, lse, n_total, B, n_tiles);
    return check_launch("attn_fwd");
}

int u3d_attn_varlen_bwd(const float* qkv, const float* out, const float* dout, const float* lse, const int32_t* cu_seqlens,
                        int B, int max_len, int64_t n_total, int H, int hd, float scale, float* dqkv, float* delta_ws,
                        double flops_hint, u3d_stream_t stream) {
    if (!qkv || !out || !dout || !lse || !cu_seqlens || !dqkv || !delta_ws || B <= 0 || H <= 0 || n_total <= 0) return U3D_EINVAL;
    if (hd != 32) { set_error("attn: head_dim %d unsupported (32 only)", hd); return U3D_EUNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_ATTN_BWD, s, flops_hint);
    if (max_len <= 0) return U3D_OK;
    if (fp32_x3()) {
        attn_bwd_x3_launch(qkv, out, dout, lse, cu_seqlens, B, max_len, n_total, H, scale, dqkv, delta_ws, s, 3);
        return check_launch("attn_bwd_x3");
    }
    hipLaunchKernelGGL(attn_delta_k, dim3((unsigned)ceil_div(n_total * H, 256)), dim3(256), 0, s, out, dout, n_total, H, delta_ws);
    const int n_tiles = (max_len + 63) / 64;
    const dim3 grid(attn_grid(H, B, n_tiles));
    hipLaunchKernelGGL(attn_bwd_dq_k, grid, dim3(256), 0, s, qkv, dout, lse, (const float*)delta_ws, cu_seqlens, H, scale, dqkv, n_total, B, n_tiles);
    hipLaunchKernelGGL(attn_bwd_dkv_k, grid, dim3(256), 0, s, qkv, dout, lse, (const float*)delta_ws, cu_seqlens, H, scale, dqkv, n_total, B, n_tiles);
    return check_launch("attn_bwd");
}

}  // extern "C"
