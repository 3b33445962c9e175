// K13 (bf16 operand form, BASELINE configs[2]): the varlen flash attention of attn.hip with bf16 MFMA operands and
// fp32 accumulation / softmax -- what `--amp` training of the reference (tools/train.py:86-99) runs through
// nn.MultiheadAttention.  qkv / out / gradients stay fp32 in HBM; tiles are rounded to bf16 (round-to-nearest-even,
// v_cvt_pk_bf16_f32) on their way into LDS / registers.
//
// v_mfma_f32_16x16x32_bf16: lane (i = lane & 15, g = lane >> 4) holds 8 consecutive reduction elements k = 8g .. 8g+7 of
// row i (A) / column i (B); the C/D fragment is the fp32 one of attn.hip (col = lane & 15, rows = 4g + r).  head_dim = 32
// is exactly one instruction's reduction depth, so
//   S^T tile (16 keys x 16 queries) = K_tile . Q^T           ONE MFMA (A = K rows from LDS, B = own Q rows in registers)
//   O (16 queries x 16 dims)       += P (16 x 32 keys) . V   ONE MFMA per 32-key block: the A operand is the pair of C
//       fragments of two 16-key tiles -- lane (query, g) holds keys {4g..4g+3} of both, which fixes the k <-> key map
//       k = 8g + e  <->  key 32t + 16 (e >> 2) + 4g + (e & 3); the B operand reads V with the same map as eight 2-byte LDS
//       loads (column access of the natural [key][dim] tile; a lane group reads 32 contiguous bytes, groups 4 rows apart).
// The backward kernels use the same two shapes in both orientations (see attn.hip's header).  With the matrix work 16x
// cheaper than in fp32 the kernels are bound by staging, LDS reads and the fp32 softmax arithmetic.
#include "u3d_common.h"

namespace u3d {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

constexpr float B_LOG2E = 1.44269504088896340736f, B_LN2 = 0.69314718055994530942f;
constexpr int HLD = 40;      // halves per staged row: 32 + 8 pad (80-byte rows keep the 16-byte fragment reads of 16 rows on distinct banks)
#define U3D_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// stage 64 rows x 32 floats of `base` into dst[64][HLD] as bf16; rows >= len are zero; optional scale before rounding
__device__ __forceinline__ void stage_tile_bf16(const float* __restrict__ base, int ld, int row0, int len, float scale, __bf16* dst, int tid) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = tid + j * 256;
        const int r = idx >> 3, c4 = idx & 7;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < len) v = *reinterpret_cast<const float4*>(base + (int64_t)(row0 + r) * ld + c4 * 4);
        const bf16x4 h = {(__bf16)(v.x * scale), (__bf16)(v.y * scale), (__bf16)(v.z * scale), (__bf16)(v.w * scale)};
        *reinterpret_cast<bf16x4*>(dst + r * HLD + c4 * 4) = h;
    }
}

// own row -> B operand: 8 consecutive dims 8g .. 8g+7 of row `p` (nullptr: zeros), scaled, rounded
__device__ __forceinline__ bf16x8 row_frag(const float* p, float scale) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (p) { a = *reinterpret_cast<const float4*>(p); b = *reinterpret_cast<const float4*>(p + 4); }
    return bf16x8{(__bf16)(a.x * scale), (__bf16)(a.y * scale), (__bf16)(a.z * scale), (__bf16)(a.w * scale),
                  (__bf16)(b.x * scale), (__bf16)(b.y * scale), (__bf16)(b.z * scale), (__bf16)(b.w * scale)};
}

// A operand from two C fragments (fp32 -> bf16): k = 8g + e <-> row 16 (e >> 2) + 4g + (e & 3) of the 32-row block
__device__ __forceinline__ bf16x8 pair_frag(const float (&lo)[4], const float (&hi)[4]) {
    return bf16x8{(__bf16)lo[0], (__bf16)lo[1], (__bf16)lo[2], (__bf16)lo[3], (__bf16)hi[0], (__bf16)hi[1], (__bf16)hi[2], (__bf16)hi[3]};
}

// B operand with the same k <-> row map: column `col` of the natural [row][dim] tile `t` (32-row block starting at row0)
__device__ __forceinline__ bf16x8 col_frag(const __bf16* t, int row0, int g, int col) {
    const __bf16* p = t + (row0 + 4 * g) * HLD + col;
    return bf16x8{p[0], p[HLD], p[2 * HLD], p[3 * HLD], p[16 * HLD], p[17 * HLD], p[18 * HLD], p[19 * HLD]};
}

struct AttnWorkB { int b, h, tile; };
__device__ __forceinline__ AttnWorkB attn_decode_b(int H, int B, int n_tiles) {      // (scene, head) -> XCD, see attn.hip
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int hb = (j / n_tiles) * 8 + x;
    AttnWorkB w;
    w.tile = j % n_tiles;
    w.h = hb % H;
    w.b = hb / H;
    return w;
}

__global__ __launch_bounds__(256) void attn_fwd_bf16_k(const float* __restrict__ qkv, const int32_t* __restrict__ cu, int H, float scale,
                                                       float* __restrict__ out, float* __restrict__ lse, int64_t n_total, int B, int n_tiles) {
    __shared__ __attribute__((aligned(16))) __bf16 Ks[64 * HLD];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[64 * HLD];
    const AttnWorkB wk_ = attn_decode_b(H, B, n_tiles);
    const int b = wk_.b, h = wk_.h;
    if (b >= B) return;
    const int start = cu[b], len = cu[b + 1] - start;
    const int q0 = wk_.tile * 64;
    if (q0 >= len) return;
    const int D = H * 32, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const float* base = qkv + (int64_t)start * ld + h * 32;
    const int qrow = q0 + wave * 16 + i16;
    // scores in log2 units: q carries scale * log2(e)
    const bf16x8 qf = row_frag(qrow < len ? base + (int64_t)qrow * ld + g * 8 : nullptr, scale * B_LOG2E);
    float m = -INFINITY, l = 0.f;
    f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int ntiles = (len + 63) >> 6;
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
        stage_tile_bf16(base + D, ld, kt * 64, len, 1.f, Ks, tid);
        stage_tile_bf16(base + 2 * D, ld, kt * 64, len, 1.f, Vs, tid);
        __syncthreads();
        float st[4][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ks + (kb * 16 + i16) * HLD + g * 8);
            const f32x4 s4 = U3D_MFMA_BF16(a, qf, (f32x4{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
            for (int r = 0; r < 4; ++r) st[kb][r] = s4[r];
        }
        if (kt == ntiles - 1 && (len & 63)) {          // only the last tile can hold keys past the end (wave-uniform)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * 64 + kb * 16 + g * 4 + r >= len) st[kb][r] = -INFINITY;
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
            const float ar = __shfl(alpha, g * 4 + r, 64);
            o[0][r] *= ar;
            o[1][r] *= ar;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 pa = pair_frag(st[2 * t], st[2 * t + 1]);
            o[0] = U3D_MFMA_BF16(pa, col_frag(Vs, 32 * t, g, i16), o[0]);
            o[1] = U3D_MFMA_BF16(pa, col_frag(Vs, 32 * t, g, 16 + i16), o[1]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float lr = __shfl(l, g * 4 + r, 64);
        const int row = q0 + wave * 16 + g * 4 + r;
        if (row < len) {
            const float inv = 1.f / lr;
            float* op = out + (int64_t)(start + row) * D + h * 32 + i16;
            op[0] = o[0][r] * inv;
            op[16] = o[1][r] * inv;
        }
    }
    if (g == 0 && qrow < len) lse[(int64_t)h * n_total + start + qrow] = m * B_LN2 + __logf(l);      // natural-log units
}

__global__ __launch_bounds__(256) void attn_delta_bf16_k(const float* __restrict__ o, const float* __restrict__ dout, int64_t n, int H, float* delta) {
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

__global__ __launch_bounds__(256) void attn_bwd_dq_bf16_k(const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ lse,
                                                          const float* __restrict__ delta, const int32_t* __restrict__ cu, int H, float scale,
                                                          float* __restrict__ dqkv, int64_t n_total, int B, int n_tiles) {
    __shared__ __attribute__((aligned(16))) __bf16 Ks[64 * HLD];
    __shared__ __attribute__((aligned(16))) __bf16 Vs[64 * HLD];
    const AttnWorkB wk_ = attn_decode_b(H, B, n_tiles);
    const int b = wk_.b, h = wk_.h;
    if (b >= B) return;
    const int start = cu[b], len = cu[b + 1] - start;
    const int q0 = wk_.tile * 64;
    if (q0 >= len) return;
    const int D = H * 32, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const float* base = qkv + (int64_t)start * ld + h * 32;
    const int qrow = q0 + wave * 16 + i16;
    const bool qok = qrow < len;
    const bf16x8 qf = row_frag(qok ? base + (int64_t)qrow * ld + g * 8 : nullptr, scale * B_LOG2E);
    const bf16x8 dof = row_frag(qok ? dout + (int64_t)(start + qrow) * D + h * 32 + g * 8 : nullptr, 1.f);
    // log2 units; rows past the end get +inf so that exp2(s - lse) = 0 masks them without a select per element
    const float lse_q = qok ? lse[(int64_t)h * n_total + start + qrow] * B_LOG2E : INFINITY;
    const float del_q = qok ? delta[(int64_t)h * n_total + start + qrow] : 0.f;
    f32x4 dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int ntiles = (len + 63) >> 6;
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
        stage_tile_bf16(base + D, ld, kt * 64, len, 1.f, Ks, tid);
        stage_tile_bf16(base + 2 * D, ld, kt * 64, len, 1.f, Vs, tid);
        __syncthreads();
        const bool last = kt == ntiles - 1 && (len & 63);          // wave-uniform
        float ds[4][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const bf16x8 ak = *reinterpret_cast<const bf16x8*>(Ks + (kb * 16 + i16) * HLD + g * 8);
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(Vs + (kb * 16 + i16) * HLD + g * 8);
            const f32x4 s4 = U3D_MFMA_BF16(ak, qf, (f32x4{0.f, 0.f, 0.f, 0.f}));
            const f32x4 dp4 = U3D_MFMA_BF16(av, dof, (f32x4{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __builtin_amdgcn_exp2f(s4[r] - lse_q);
                if (last && kt * 64 + kb * 16 + g * 4 + r >= len) p = 0.f;        // zero-padded keys of the last tile
                ds[kb][r] = p * (dp4[r] - del_q);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 da = pair_frag(ds[2 * t], ds[2 * t + 1]);
            dq[0] = U3D_MFMA_BF16(da, col_frag(Ks, 32 * t, g, i16), dq[0]);
            dq[1] = U3D_MFMA_BF16(da, col_frag(Ks, 32 * t, g, 16 + i16), dq[1]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = q0 + wave * 16 + g * 4 + r;
        if (row < len) {
            float* op = dqkv + (int64_t)(start + row) * ld + h * 32 + i16;
            op[0] = dq[0][r] * scale;
            op[16] = dq[1][r] * scale;
        }
    }
}

__global__ __launch_bounds__(256) void attn_bwd_dkv_bf16_k(const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ lse,
                                                           const float* __restrict__ delta, const int32_t* __restrict__ cu, int H, float scale,
                                                           float* __restrict__ dqkv, int64_t n_total, int B, int n_tiles) {
    __shared__ __attribute__((aligned(16))) __bf16 Qs[64 * HLD];
    __shared__ __attribute__((aligned(16))) __bf16 Os[64 * HLD];
    __shared__ float lse_s[64], del_s[64];
    const AttnWorkB wk_ = attn_decode_b(H, B, n_tiles);
    const int b = wk_.b, h = wk_.h;
    if (b >= B) return;
    const int start = cu[b], len = cu[b + 1] - start;
    const int k0 = wk_.tile * 64;
    if (k0 >= len) return;
    const int D = H * 32, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const float* base = qkv + (int64_t)start * ld + h * 32;
    const float* dobase = dout + (int64_t)start * D + h * 32;
    const int krow = k0 + wave * 16 + i16;
    const bf16x8 kf = row_frag(krow < len ? base + (int64_t)krow * ld + D + g * 8 : nullptr, 1.f);
    const bf16x8 vf = row_frag(krow < len ? base + (int64_t)krow * ld + 2 * D + g * 8 : nullptr, 1.f);
    f32x4 dk[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int ntiles = (len + 63) >> 6;
    for (int qt = 0; qt < ntiles; ++qt) {
        __syncthreads();
        stage_tile_bf16(base, ld, qt * 64, len, scale * B_LOG2E, Qs, tid);        // log2 units; dK is rescaled by ln 2 at the end
        stage_tile_bf16(dobase, D, qt * 64, len, 1.f, Os, tid);
        if (tid < 64) {
            const int q = qt * 64 + tid;
            lse_s[tid] = q < len ? lse[(int64_t)h * n_total + start + q] * B_LOG2E : INFINITY;   // exp2(s - inf) = 0 masks the row
            del_s[tid] = q < len ? delta[(int64_t)h * n_total + start + q] : 0.f;
        }
        __syncthreads();
        float p[4][4], ds[4][4];
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            const bf16x8 aq = *reinterpret_cast<const bf16x8*>(Qs + (qb * 16 + i16) * HLD + g * 8);
            const bf16x8 ao = *reinterpret_cast<const bf16x8*>(Os + (qb * 16 + i16) * HLD + g * 8);
            const f32x4 s4 = U3D_MFMA_BF16(aq, kf, (f32x4{0.f, 0.f, 0.f, 0.f}));
            const f32x4 dp4 = U3D_MFMA_BF16(ao, vf, (f32x4{0.f, 0.f, 0.f, 0.f}));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qq = qb * 16 + g * 4 + r;
                p[qb][r] = __builtin_amdgcn_exp2f(s4[r] - lse_s[qq]);
                ds[qb][r] = p[qb][r] * (dp4[r] - del_s[qq]);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 pa = pair_frag(p[2 * t], p[2 * t + 1]);
            const bf16x8 da = pair_frag(ds[2 * t], ds[2 * t + 1]);
            dv[0] = U3D_MFMA_BF16(pa, col_frag(Os, 32 * t, g, i16), dv[0]);
            dv[1] = U3D_MFMA_BF16(pa, col_frag(Os, 32 * t, g, 16 + i16), dv[1]);
            dk[0] = U3D_MFMA_BF16(da, col_frag(Qs, 32 * t, g, i16), dk[0]);
            dk[1] = U3D_MFMA_BF16(da, col_frag(Qs, 32 * t, g, 16 + i16), dk[1]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = k0 + wave * 16 + g * 4 + r;
        if (row < len) {
            float* op = dqkv + (int64_t)(start + row) * ld + h * 32 + i16;
            op[D] = dk[0][r] * B_LN2;
            op[D + 16] = dk[1][r] * B_LN2;
            op[2 * D] = dv[0][r];
            op[2 * D + 16] = dv[1][r];
        }
    }
}

}  // namespace u3d

using namespace u3d;

extern "C" {

int u3d_attn_varlen_fwd_bf16(const float* qkv, const int32_t* cu_seqlens, int B, int max_len, int64_t n_total, int H, int hd,
                             float scale, float* out, float* lse, double flops_hint, u3d_stream_t stream) {
    if (!qkv || !cu_seqlens || !out || !lse || B <= 0 || H <= 0 || n_total <= 0) return U3D_EINVAL;
    if (hd != 32) { set_error("attn: head_dim %d unsupported (32 only)", hd); return U3D_EUNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_ATTN_FWD, s, flops_hint);
    if (max_len <= 0) return U3D_OK;
    const int n_tiles = (max_len + 63) / 64;
    const unsigned grid = (unsigned)(((H * B + 7) / 8) * 8 * n_tiles);
    hipLaunchKernelGGL(attn_fwd_bf16_k, dim3(grid), dim3(256), 0, s, qkv, cu_seqlens, H, scale, out, lse, n_total, B, n_tiles);
    return check_launch("attn_fwd_bf16");
}

int u3d_attn_varlen_bwd_bf16(const float* qkv, const float* out, const float* dout, const float* lse, const int32_t* cu_seqlens,
                             int B, int max_len, int64_t n_total, int H, int hd, float scale, float* dqkv, float* delta_ws,
                             double flops_hint, u3d_stream_t stream) {
    if (!qkv || !out || !dout || !lse || !cu_seqlens || !dqkv || !delta_ws || B <= 0 || H <= 0 || n_total <= 0) return U3D_EINVAL;
    if (hd != 32) { set_error("attn: head_dim %d unsupported (32 only)", hd); return U3D_EUNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_ATTN_BWD, s, flops_hint);
    if (max_len <= 0) return U3D_OK;
    hipLaunchKernelGGL(attn_delta_bf16_k, dim3((unsigned)ceil_div(n_total * H, 256)), dim3(256), 0, s, out, dout, n_total, H, delta_ws);
    const int n_tiles = (max_len + 63) / 64;
    const dim3 grid((unsigned)(((H * B + 7) / 8) * 8 * n_tiles));
    hipLaunchKernelGGL(attn_bwd_dq_bf16_k, grid, dim3(256), 0, s, qkv, dout, lse, (const float*)delta_ws, cu_seqlens, H, scale, dqkv, n_total, B, n_tiles);
    hipLaunchKernelGGL(attn_bwd_dkv_bf16_k, grid, dim3(256), 0, s, qkv, dout, lse, (const float*)delta_ws, cu_seqlens, H, scale, dqkv, n_total, B, n_tiles);
    return check_launch("attn_bwd_bf16");
}

}  // extern "C"
