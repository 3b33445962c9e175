// K13 (fp32 math on the bf16 matrix pipe, u3d_common.h "bf16x3"): the varlen flash attention of attn.hip with every fp32
// operand split exactly into three bf16 planes and every product formed from six v_mfma_f32_16x16x32_bf16 -- the error of an
// fp32 FMA chain at 6 x 16 instead of 8 x 32 matrix-pipe cycles per 16 x 16 x 32 block.  The default path of
// u3d_attn_varlen_fwd / _bwd (reference: nn.MultiheadAttention inside unidet3d/encoder.py:19-21,55-61; oracle/model.py);
// U3D_FP32_MATH=mfma selects the native fp32 MFMA kernels of attn.hip.
//
// v_mfma_f32_16x16x32_bf16: lane (i = lane & 15, g = lane >> 4) holds the 8 reduction elements k = 8g .. 8g+7 of row i (A) /
// column i (B); head_dim = 32 is one instruction's reduction depth.
//   S^T (16 keys x 16 queries) = K_tile . Q^T       A = K rows, natural [key][dim] planes in LDS (one 16-byte read per plane),
//                                                   B = own Q rows, split once into registers
//   O (16 queries x 16 dims) += P . V               A = two C fragments (keys {4g..4g+3} of two 16-key tiles), split in registers,
//                                                   B = a COLUMN of the natural V planes: the fragment k = 8g + e <-> key
//       32t + 16 (e >> 2) + 4g + (e & 3) of dim i is two ds_read_b64_tr_b16 per plane -- a 16-lane group hands in the [4 keys][16 dims]
//       block of its lane group (lane t the 8 bytes at key t >> 2, dims 4 (t & 3) ..) and lane t receives dim t over the four keys.
//       (Rounds 2-3 staged such tensors a second time in a "pair" layout -- two keys per dword, [dim][key-pair slot] -- read with
//       ds_read_b128; the transpose read makes that copy unnecessary: half the LDS per staged tensor, no second packing and no
//       dword stores in the staging threads.  Rows 4g + j and 4 (g + 1) + j of a plane differ by 2 in the chunk XOR, so the 32
//       lanes of a read cover 256 consecutive-bank bytes: conflict-free.)
// The backward kernels read their staged tiles in both orientations the same way: rows with ds_read_b128, columns with transpose reads.
// The kernels are templates over the number of planes: NP = 3 is the fp32 path above, NP = 1 the bf16-operand form of
// BASELINE configs[2] (one bf16 value per operand, rounded to nearest even; u3d_attn_varlen_*_bf16 at the end of the file).
#include <mutex>
#include <stdlib.h>

#include "u3d_common.h"

namespace u3d {

typedef bf16x8_t bf16x8;

constexpr float X_LOG2E = 1.44269504088896340736f, X_LN2 = 0.69314718055994530942f;
constexpr int XLD = 32;                  // halves per row of a natural plane: unpadded, the 16-byte chunk index is XORed with (row >> 1) & 3
                                         // (conflict-free ds_read_b128 for the lane -> (row = lane & 15, chunk = lane >> 4) map)
constexpr int XN = 64 * XLD;             // halves per natural plane
#define U3D_MFMA_X(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// c += a . b with both operands in three planes takes h.l, m.m, l.h, h.m, m.h, h.h (smallest terms first);
// two independent accumulators side by side (dependent MFMAs wait for their predecessor; alternating chains fills the gaps).
// A bf16 MFMA truncates its 32 products at the exponent of its C operand, always towards zero: plane products 2^-8 .. 2^-16 below
// a running sum lose their low bits every time -- a coherent bias (tools/bias_probe.py: -1.2e-8 mean error after 8 blocks, zero
// for fp32 MFMAs) that reductions over thousands of rows downstream do not average out.  So chains that start from zero (S, dP)
// run smallest terms first, and the running sums O / dQ / dK / dV keep the low-order products in accumulators of their own.
template <int NP>
__device__ __forceinline__ void mfma_x3_2a(const bf16x8 (&a0)[NP], const bf16x8 (&a1)[NP], const bf16x8 (&b)[NP], f32x4& c0, f32x4& c1) {
#pragma unroll
    for (int o = NP - 1; o >= 0; --o)
#pragma unroll
        for (int qa = 0; qa <= o; ++qa) {
            c0 = U3D_MFMA_X(a0[qa], b[o - qa], c0);
            c1 = U3D_MFMA_X(a1[qa], b[o - qa], c1);
        }
}
// running sums: the h.h product goes to (c0, c1), the five low-order plane products to their own accumulators (l0, l1), joined
// once at the end of the kernel (NP = 1, bf16 operands: the one product goes to (c0, c1))
template <int NP>
__device__ __forceinline__ void mfma_x3_2b(const bf16x8 (&a)[NP], const bf16x8 (&b0)[NP], const bf16x8 (&b1)[NP], f32x4& c0, f32x4& c1,
                                           f32x4& l0, f32x4& l1) {
#pragma unroll
    for (int o = NP - 1; o >= 1; --o)
#pragma unroll
        for (int qa = 0; qa <= o; ++qa) {
            l0 = U3D_MFMA_X(a[qa], b0[o - qa], l0);
            l1 = U3D_MFMA_X(a[qa], b1[o - qa], l1);
        }
    c0 = U3D_MFMA_X(a[0], b0[0], c0);
    c1 = U3D_MFMA_X(a[0], b1[0], c1);
}
// one chain pair sharing nothing: s += a . b, d += c . e (S and dP of the backward kernels)
template <int NP>
__device__ __forceinline__ void mfma_x3_2c(const bf16x8 (&a)[NP], const bf16x8 (&b)[NP], const bf16x8 (&c)[NP], const bf16x8 (&e)[NP], f32x4& s, f32x4& d) {
#pragma unroll
    for (int o = NP - 1; o >= 0; --o)
#pragma unroll
        for (int qa = 0; qa <= o; ++qa) {
            s = U3D_MFMA_X(a[qa], b[o - qa], s);
            d = U3D_MFMA_X(c[qa], e[o - qa], d);
        }
}

// NP planes of a pair / of eight values: NP = 3 the exact split (u3d_common.h), NP = 1 one bf16 value rounded to nearest even
// (bf16 operands, BASELINE configs[2])
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_r;
template <int NP>
__device__ __forceinline__ void planes_pair(float a, float b, unsigned (&w)[NP]) {
    if constexpr (NP == 3) split3_pair(a, b, w[0], w[1], w[2]);
    else w[0] = __builtin_bit_cast(unsigned, bf16x2_r{(__bf16)a, (__bf16)b});
}
template <int NP>
__device__ __forceinline__ void planes_x8(const f32x4& lo, const f32x4& hi, bf16x8 (&out)[NP]) {
    if constexpr (NP == 3) split3_x8(lo, hi, out);
    else out[0] = bf16x8{(__bf16)lo[0], (__bf16)lo[1], (__bf16)lo[2], (__bf16)lo[3], (__bf16)hi[0], (__bf16)hi[1], (__bf16)hi[2], (__bf16)hi[3]};
}

// Stage 64 rows x 32 floats of `base` (rows >= len are zero, values scaled before the split): thread (p = tid >> 3, c = tid & 7)
// holds dims 4c .. 4c+3 of rows 2p and 2p+1; load_x3 issues the global loads (one tile ahead of their use: the compute phase of
// the tile before covers their latency), store_x3 splits and writes the natural planes nat[NP][64][XLD] halves.
struct StageRegs { f32x4 a, b; };
__device__ __forceinline__ StageRegs load_x3(const float* __restrict__ base, int ld, int row0, int len, int tid) {
    const int p = tid >> 3, c = tid & 7;
    const int r0 = row0 + 2 * p;
    StageRegs r;
    r.a = f32x4{0.f, 0.f, 0.f, 0.f};
    r.b = r.a;
    if (r0 < len) r.a = *reinterpret_cast<const f32x4*>(base + (int64_t)r0 * ld + c * 4);
    if (r0 + 1 < len) r.b = *reinterpret_cast<const f32x4*>(base + (int64_t)(r0 + 1) * ld + c * 4);
    return r;
}
template <int NP>
__device__ __forceinline__ void store_x3(StageRegs r, float scale, __bf16* nat, int tid) {
    const int p = tid >> 3, c = tid & 7;
    const f32x4 a = r.a * scale, b = r.b * scale;
    unsigned wa[2][NP], wb[2][NP];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        planes_pair<NP>(a[2 * j], a[2 * j + 1], wa[j]);
        planes_pair<NP>(b[2 * j], b[2 * j + 1], wb[j]);
    }
    const int off = (((c >> 1) ^ (p & 3)) * 8) + (c & 1) * 4;          // rows 2p and 2p+1 share (row >> 1) & 3 = p & 3
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        *reinterpret_cast<uint2*>(nat + q * XN + (2 * p) * XLD + off) = make_uint2(wa[0][q], wa[1][q]);
        *reinterpret_cast<uint2*>(nat + q * XN + (2 * p + 1) * XLD + off) = make_uint2(wb[0][q], wb[1][q]);
    }
}

// bf16 tensors in HBM (IO16; one plane, include/u3d.h u3d_attn_varlen_*_b16): the same staging map with 8-byte loads and no arithmetic --
// the staged values are the tensor's own, so a scale cannot be folded in here (the kernels apply it to the scores / to dK instead)
struct StageRegs16 { uint2 a, b; };
__device__ __forceinline__ StageRegs16 load_x16(const __bf16* __restrict__ base, int ld, int row0, int len, int tid) {
    const int p = tid >> 3, c = tid & 7;
    const int r0 = row0 + 2 * p;
    StageRegs16 r;
    r.a = make_uint2(0u, 0u);
    r.b = r.a;
    if (r0 < len) r.a = *reinterpret_cast<const uint2*>(base + (int64_t)r0 * ld + c * 4);
    if (r0 + 1 < len) r.b = *reinterpret_cast<const uint2*>(base + (int64_t)(r0 + 1) * ld + c * 4);
    return r;
}
__device__ __forceinline__ void store_x16(StageRegs16 r, __bf16* nat, int tid) {
    const int p = tid >> 3, c = tid & 7;
    const int off = (((c >> 1) ^ (p & 3)) * 8) + (c & 1) * 4;
    *reinterpret_cast<uint2*>(nat + (2 * p) * XLD + off) = r.a;
    *reinterpret_cast<uint2*>(nat + (2 * p + 1) * XLD + off) = r.b;
}
__device__ __forceinline__ void row_frag_x16(const __bf16* ptr, bf16x8 (&out)[1]) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ptr) v = *reinterpret_cast<const u32x4*>(ptr);
    out[0] = __builtin_bit_cast(bf16x8, v);
}
template <bool IO16> struct IoT { typedef float t; typedef StageRegs regs; };
template <> struct IoT<true> { typedef __bf16 t; typedef StageRegs16 regs; };
template <bool IO16>
__device__ __forceinline__ typename IoT<IO16>::regs load_io(const typename IoT<IO16>::t* __restrict__ base, int ld, int row0, int len, int tid) {
    if constexpr (IO16) return load_x16(base, ld, row0, len, tid);
    else return load_x3(base, ld, row0, len, tid);
}
template <int NP, bool IO16>
__device__ __forceinline__ void store_io(typename IoT<IO16>::regs r, float scale, __bf16* nat, int tid) {
    if constexpr (IO16) store_x16(r, nat, tid);
    else store_x3<NP>(r, scale, nat, tid);
}
template <bool IO16>
__device__ __forceinline__ void put_io(typename IoT<IO16>::t* p, float v) {
    if constexpr (IO16) *p = (__bf16)v;
    else *p = v;
}

// own row -> B operand planes: 8 consecutive dims of row `ptr` (nullptr: zeros), scaled
template <int NP>
__device__ __forceinline__ void row_frag_x3(const float* ptr, float scale, bf16x8 (&out)[NP]) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
    if (ptr) { a = *reinterpret_cast<const f32x4*>(ptr); b = *reinterpret_cast<const f32x4*>(ptr + 4); }
    a *= scale;
    b *= scale;
    planes_x8<NP>(a, b, out);
}

// A operand planes from two C fragments: k = 8g + e <-> row 16 (e >> 2) + 4g + (e & 3) of the 32-row block
template <int NP>
__device__ __forceinline__ void pair_frag_x3(const float (&lo)[4], const float (&hi)[4], bf16x8 (&out)[NP]) {
    planes_x8<NP>(f32x4{lo[0], lo[1], lo[2], lo[3]}, f32x4{hi[0], hi[1], hi[2], hi[3]}, out);
}

// natural planes: rows 16 kb + i, dims 8g .. 8g+7
template <int NP>
__device__ __forceinline__ void nat_frag_x3(const __bf16* nat, int kb, int i16, int g, bf16x8 (&out)[NP]) {
#pragma unroll
    for (int q = 0; q < NP; ++q) out[q] = *reinterpret_cast<const bf16x8*>(nat + q * XN + (kb * 16 + i16) * XLD + ((g ^ ((i16 >> 1) & 3)) * 8));
}

// natural planes read by COLUMN: dim 16 cb + i16 over rows {4g..4g+3} and {16+4g..16+4g+3} of the 32-row block t (the k order of
// pair_frag_x3).  ds_read_b64_tr_b16: lane i16 of a 16-lane group passes the address of 4 halves -- row (i16 >> 2) of the group's four,
// dims 16 cb + 4 (i16 & 3) .. -- and receives dim 16 cb + i16 of the four rows.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
template <int NP>
__device__ __forceinline__ void tr_col_frag_x3(const __bf16* nat, int t, int g, int i16, int cb, bf16x8 (&out)[NP]) {
    const int row = 32 * t + 4 * g + (i16 >> 2), pc = i16 & 3;
    const __bf16* s = nat + row * XLD + (((2 * cb + (pc >> 1)) ^ ((row >> 1) & 3)) * 8) + 4 * (pc & 1);      // row + 16 has the same chunk XOR
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const s16x4_t r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(s + q * XN));
        const s16x4_t r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(s + q * XN + 16 * XLD));
        out[q] = __builtin_bit_cast(bf16x8, s16x8_t{r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]});
    }
}

struct AttnWorkX { int b, h, tile; };
__device__ __forceinline__ AttnWorkX attn_decode_x(int H, int B, int n_tiles) {      // (scene, head) -> XCD, see attn.hip
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int hb = (j / n_tiles) * 8 + x;
    AttnWorkX w;
    w.tile = j % n_tiles;
    w.h = hb % H;
    w.b = hb / H;
    return w;
}

template <int NP, bool IO16 = false>
__global__ __launch_bounds__(256) void attn_fwd_x3_k(const typename IoT<IO16>::t* __restrict__ qkv, const int32_t* __restrict__ cu, int H, float scale,
                                                     typename IoT<IO16>::t* __restrict__ out, float* __restrict__ lse, int64_t n_total, int B, int n_tiles) {
    static_assert(!IO16 || NP == 1, "bf16 tensors carry one plane");
    typedef typename IoT<IO16>::t io_t;
    __shared__ __attribute__((aligned(16))) __bf16 Kn[NP * XN];
    __shared__ __attribute__((aligned(16))) __bf16 Vn[NP * XN];
    const AttnWorkX wk_ = attn_decode_x(H, B, n_tiles);
    const int b = wk_.b, h = wk_.h;
    if (b >= B) return;
    const int start = cu[b], len = cu[b + 1] - start;
    const int q0 = wk_.tile * 64;
    if (q0 >= len) return;
    const int D = H * 32, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const io_t* base = qkv + (int64_t)start * ld + h * 32;
    const int qrow = q0 + wave * 16 + i16;
    bf16x8 qf[NP];                                // scores in log2 units: q carries scale * log2(e) -- or, IO16, the scores are scaled
    if constexpr (IO16) row_frag_x16(qrow < len ? base + (int64_t)qrow * ld + g * 8 : nullptr, qf);
    else row_frag_x3(qrow < len ? base + (int64_t)qrow * ld + g * 8 : nullptr, scale * X_LOG2E, qf);
    const float sc_ = IO16 ? scale * X_LOG2E : 1.f;
    float m = -INFINITY, l = 0.f;
    f32x4 o[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, ol[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // h.h | low-order products
    const int ntiles = (len + 63) >> 6;
    typename IoT<IO16>::regs rk = load_io<IO16>(base + D, ld, 0, len, tid), rv = load_io<IO16>(base + 2 * D, ld, 0, len, tid);
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
        store_io<NP, IO16>(rk, 1.f, Kn, tid);
        store_io<NP, IO16>(rv, 1.f, Vn, tid);
        if (kt + 1 < ntiles) {
            rk = load_io<IO16>(base + D, ld, kt * 64 + 64, len, tid);
            rv = load_io<IO16>(base + 2 * D, ld, kt * 64 + 64, len, tid);
        }
        __syncthreads();
        float st[4][4];
#pragma unroll
        for (int kb = 0; kb < 4; kb += 2) {
            bf16x8 a0[NP], a1[NP];
            nat_frag_x3(Kn, kb, i16, g, a0);
            nat_frag_x3(Kn, kb + 1, i16, g, a1);
            f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
            mfma_x3_2a(a0, a1, qf, s0, s1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (IO16) { st[kb][r] = s0[r] * sc_; st[kb + 1][r] = s1[r] * sc_; }
                else { st[kb][r] = s0[r]; st[kb + 1][r] = s1[r]; }
            }
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
            ol[0][r] *= ar;
            ol[1][r] *= ar;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bf16x8 pa[NP], v0[NP], v1[NP];
            pair_frag_x3(st[2 * t], st[2 * t + 1], pa);
            tr_col_frag_x3(Vn, t, g, i16, 0, v0);
            tr_col_frag_x3(Vn, t, g, i16, 1, v1);
            mfma_x3_2b(pa, v0, v1, o[0], o[1], ol[0], ol[1]);
        }
    }
    o[0] += ol[0];
    o[1] += ol[1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float lr = __shfl(l, g * 4 + r, 64);
        const int row = q0 + wave * 16 + g * 4 + r;
        if (row < len) {
            const float inv = 1.f / lr;
            io_t* op = out + (int64_t)(start + row) * D + h * 32 + i16;
            put_io<IO16>(op, o[0][r] * inv);
            put_io<IO16>(op + 16, o[1][r] * inv);
        }
    }
    if (g == 0 && qrow < len) lse[(int64_t)h * n_total + start + qrow] = m * X_LN2 + __logf(l);      // natural-log units
}

__global__ __launch_bounds__(256) void attn_delta_x16_k(const __bf16* __restrict__ o, const __bf16* __restrict__ dout, int64_t n, int H, float* delta) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * H) return;
    const int64_t i = idx / H;
    const int h = (int)(idx % H);
    const u32x4* a = reinterpret_cast<const u32x4*>(o + i * H * 32 + h * 32);
    const u32x4* b = reinterpret_cast<const u32x4*>(dout + i * H * 32 + h * 32);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const u32x4 x = a[j], y = b[j];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            s += __builtin_bit_cast(float, x[c] << 16) * __builtin_bit_cast(float, y[c] << 16) +
                 __builtin_bit_cast(float, x[c] & 0xffff0000u) * __builtin_bit_cast(float, y[c] & 0xffff0000u);
    }
    delta[(int64_t)h * n + i] = s;
}

__global__ __launch_bounds__(256) void attn_delta_x3_k(const float* __restrict__ o, const float* __restrict__ dout, int64_t n, int H, float* delta) {
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

// dQ: one workgroup per 64-query tile, keys streamed.  K is read by rows for S and by columns for dQ += dS . K.
template <int NP, bool IO16 = false>
__global__ __launch_bounds__(256) void attn_bwd_dq_x3_k(const typename IoT<IO16>::t* __restrict__ qkv, const typename IoT<IO16>::t* __restrict__ dout, const float* __restrict__ lse,
                                                        const float* __restrict__ delta, const int32_t* __restrict__ cu, int H, float scale,
                                                        typename IoT<IO16>::t* __restrict__ dqkv, int64_t n_total, int B, int n_tiles) {
    static_assert(!IO16 || NP == 1, "bf16 tensors carry one plane");
    typedef typename IoT<IO16>::t io_t;
    __shared__ __attribute__((aligned(16))) __bf16 Kn[NP * XN];
    __shared__ __attribute__((aligned(16))) __bf16 Vn[NP * XN];
    const AttnWorkX wk_ = attn_decode_x(H, B, n_tiles);
    const int b = wk_.b, h = wk_.h;
    if (b >= B) return;
    const int start = cu[b], len = cu[b + 1] - start;
    const int q0 = wk_.tile * 64;
    if (q0 >= len) return;
    const int D = H * 32, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const io_t* base = qkv + (int64_t)start * ld + h * 32;
    const int qrow = q0 + wave * 16 + i16;
    const bool qok = qrow < len;
    bf16x8 qf[NP], dof[NP];
    if constexpr (IO16) {
        row_frag_x16(qok ? base + (int64_t)qrow * ld + g * 8 : nullptr, qf);
        row_frag_x16(qok ? dout + (int64_t)(start + qrow) * D + h * 32 + g * 8 : nullptr, dof);
    } else {
        row_frag_x3(qok ? base + (int64_t)qrow * ld + g * 8 : nullptr, scale * X_LOG2E, qf);
        row_frag_x3(qok ? dout + (int64_t)(start + qrow) * D + h * 32 + g * 8 : nullptr, 1.f, dof);
    }
    const float sc_ = IO16 ? scale * X_LOG2E : 1.f;
    // log2 units; rows past the end get +inf so that exp2(s - lse) = 0 masks them without a select per element
    const float lse_q = qok ? lse[(int64_t)h * n_total + start + qrow] * X_LOG2E : INFINITY;
    const float del_q = qok ? delta[(int64_t)h * n_total + start + qrow] : 0.f;
    f32x4 dq[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dql[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int ntiles = (len + 63) >> 6;
    typename IoT<IO16>::regs rk = load_io<IO16>(base + D, ld, 0, len, tid), rv = load_io<IO16>(base + 2 * D, ld, 0, len, tid);
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();
        store_io<NP, IO16>(rk, 1.f, Kn, tid);
        store_io<NP, IO16>(rv, 1.f, Vn, tid);
        if (kt + 1 < ntiles) {
            rk = load_io<IO16>(base + D, ld, kt * 64 + 64, len, tid);
            rv = load_io<IO16>(base + 2 * D, ld, kt * 64 + 64, len, tid);
        }
        __syncthreads();
        const bool last = kt == ntiles - 1 && (len & 63);          // wave-uniform
        float ds[4][4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            bf16x8 ak[NP], av[NP];
            nat_frag_x3(Kn, kb, i16, g, ak);
            nat_frag_x3(Vn, kb, i16, g, av);
            f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, dp4 = s4;
            mfma_x3_2c<NP>(ak, qf, av, dof, s4, dp4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __builtin_amdgcn_exp2f(IO16 ? s4[r] * sc_ - lse_q : s4[r] - lse_q);
                if (last && kt * 64 + kb * 16 + g * 4 + r >= len) p = 0.f;        // zero-padded keys of the last tile
                ds[kb][r] = p * (dp4[r] - del_q);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bf16x8 da[NP], k0[NP], k1[NP];
            pair_frag_x3(ds[2 * t], ds[2 * t + 1], da);
            tr_col_frag_x3(Kn, t, g, i16, 0, k0);
            tr_col_frag_x3(Kn, t, g, i16, 1, k1);
            mfma_x3_2b(da, k0, k1, dq[0], dq[1], dql[0], dql[1]);
        }
    }
    dq[0] += dql[0];
    dq[1] += dql[1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = q0 + wave * 16 + g * 4 + r;
        if (row < len) {
            io_t* op = dqkv + (int64_t)(start + row) * ld + h * 32 + i16;
            put_io<IO16>(op, dq[0][r] * scale);
            put_io<IO16>(op + 16, dq[1][r] * scale);
        }
    }
}

// dK, dV: one workgroup per 64-key tile, queries streamed.  Q and dO are read by rows (S, dP) and by columns (dK, dV).
// (three workgroups per CU: the compiler settles at 178 VGPRs without the bound and 166, no spills, with it)
template <int NP, bool IO16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NP == 3 ? 3 : 1))) void attn_bwd_dkv_x3_k(const typename IoT<IO16>::t* __restrict__ qkv, const typename IoT<IO16>::t* __restrict__ dout, const float* __restrict__ lse,
                                                         const float* __restrict__ delta, const int32_t* __restrict__ cu, int H, float scale,
                                                         typename IoT<IO16>::t* __restrict__ dqkv, int64_t n_total, int B, int n_tiles) {
    static_assert(!IO16 || NP == 1, "bf16 tensors carry one plane");
    typedef typename IoT<IO16>::t io_t;
    __shared__ __attribute__((aligned(16))) __bf16 Qn[NP * XN];
    __shared__ __attribute__((aligned(16))) __bf16 On[NP * XN];
    __shared__ float lse_s[64], del_s[64];
    const AttnWorkX wk_ = attn_decode_x(H, B, n_tiles);
    const int b = wk_.b, h = wk_.h;
    if (b >= B) return;
    const int start = cu[b], len = cu[b + 1] - start;
    const int k0 = wk_.tile * 64;
    if (k0 >= len) return;
    const int D = H * 32, ld = 3 * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, g = lane >> 4;
    const io_t* base = qkv + (int64_t)start * ld + h * 32;
    const io_t* dobase = dout + (int64_t)start * D + h * 32;
    const int krow = k0 + wave * 16 + i16;
    bf16x8 kf[NP], vf[NP];
    if constexpr (IO16) {
        row_frag_x16(krow < len ? base + (int64_t)krow * ld + D + g * 8 : nullptr, kf);
        row_frag_x16(krow < len ? base + (int64_t)krow * ld + 2 * D + g * 8 : nullptr, vf);
    } else {
        row_frag_x3(krow < len ? base + (int64_t)krow * ld + D + g * 8 : nullptr, 1.f, kf);
        row_frag_x3(krow < len ? base + (int64_t)krow * ld + 2 * D + g * 8 : nullptr, 1.f, vf);
    }
    const float sc_ = IO16 ? scale * X_LOG2E : 1.f;
    f32x4 dk[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 dkl[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dvl[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};       // low-order plane products
    const int ntiles = (len + 63) >> 6;
    typename IoT<IO16>::regs rq = load_io<IO16>(base, ld, 0, len, tid), ro = load_io<IO16>(dobase, D, 0, len, tid);
    for (int qt = 0; qt < ntiles; ++qt) {
        __syncthreads();
        store_io<NP, IO16>(rq, scale * X_LOG2E, Qn, tid);  // log2 units; dK is rescaled by ln 2 at the end (IO16: Q as it is, scores scaled, dK by `scale`)
        store_io<NP, IO16>(ro, 1.f, On, tid);
        if (qt + 1 < ntiles) {
            rq = load_io<IO16>(base, ld, qt * 64 + 64, len, tid);
            ro = load_io<IO16>(dobase, D, qt * 64 + 64, len, tid);
        }
        if (tid < 64) {
            const int q = qt * 64 + tid;
            lse_s[tid] = q < len ? lse[(int64_t)h * n_total + start + q] * X_LOG2E : INFINITY;   // exp2(s - inf) = 0 masks the row
            del_s[tid] = q < len ? delta[(int64_t)h * n_total + start + q] : 0.f;
        }
        __syncthreads();
        #pragma unroll
        for (int t = 0; t < 2; ++t) {
            float p[2][4], ds[2][4];          // 32 queries at a time: S / dP of two 16-query blocks, then their dV / dK products
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int qb = 2 * t + u;
                bf16x8 aq[NP], ao[NP];
                nat_frag_x3(Qn, qb, i16, g, aq);
                nat_frag_x3(On, qb, i16, g, ao);
                f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, dp4 = s4;
                mfma_x3_2c<NP>(aq, kf, ao, vf, s4, dp4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = qb * 16 + g * 4 + r;
                    p[u][r] = __builtin_amdgcn_exp2f(IO16 ? s4[r] * sc_ - lse_s[qq] : s4[r] - lse_s[qq]);
                    ds[u][r] = p[u][r] * (dp4[r] - del_s[qq]);
                }
            }
            bf16x8 pa[NP], da[NP], f0[NP], f1[NP];
            pair_frag_x3(p[0], p[1], pa);
            tr_col_frag_x3(On, t, g, i16, 0, f0);
            tr_col_frag_x3(On, t, g, i16, 1, f1);
            mfma_x3_2b(pa, f0, f1, dv[0], dv[1], dvl[0], dvl[1]);
            pair_frag_x3(ds[0], ds[1], da);
            tr_col_frag_x3(Qn, t, g, i16, 0, f0);
            tr_col_frag_x3(Qn, t, g, i16, 1, f1);
            mfma_x3_2b(da, f0, f1, dk[0], dk[1], dkl[0], dkl[1]);
        }
    }
    dv[0] += dvl[0]; dv[1] += dvl[1];
    dk[0] += dkl[0]; dk[1] += dkl[1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = k0 + wave * 16 + g * 4 + r;
        if (row < len) {
            io_t* op = dqkv + (int64_t)(start + row) * ld + h * 32 + i16;
            const float ks = IO16 ? scale : X_LN2;
            put_io<IO16>(op + D, dk[0][r] * ks);
            put_io<IO16>(op + D + 16, dk[1][r] * ks);
            put_io<IO16>(op + 2 * D, dv[0][r]);
            put_io<IO16>(op + 2 * D + 16, dv[1][r]);
        }
    }
}

// launchers, called from attn.hip's entry points when fp32_x3() (arguments already validated there)
void attn_fwd_x3_launch(const float* qkv, const int32_t* cu, int B, int max_len, int64_t n_total, int H, float scale, float* out, float* lse,
                        hipStream_t s, int planes) {
    const int n_tiles = (max_len + 63) / 64;
    const unsigned grid = (unsigned)(((H * B + 7) / 8) * 8 * n_tiles);
    if (planes == 16) hipLaunchKernelGGL((attn_fwd_x3_k<1, true>), dim3(grid), dim3(256), 0, s, (const __bf16*)qkv, cu, H, scale, (__bf16*)out, lse, n_total, B, n_tiles);      // bf16 tensors
    else if (planes == 1) hipLaunchKernelGGL((attn_fwd_x3_k<1>), dim3(grid), dim3(256), 0, s, qkv, cu, H, scale, out, lse, n_total, B, n_tiles);
    else hipLaunchKernelGGL((attn_fwd_x3_k<3>), dim3(grid), dim3(256), 0, s, qkv, cu, H, scale, out, lse, n_total, B, n_tiles);
}

// dQ and dK / dV are independent given delta: with U3D_ATTN_FORK=1 the dQ kernel is forked onto a per-device side stream and joined
// before the call hands `s` back (events only; nothing blocks the host).  Each of the two grids leaves the last of its ~2.8 waves of
// workgroups a quarter full; side by side they fill each other's tail: attention backward 4.88 / 4.95 -> 4.75 / 4.76 ms per step
// alone -- and the training step got SLOWER (302.3 / 301.2 -> 297.8 / 296.8 scenes/s, same box, round 5): the fork competes with the
// weight-gradient chain that already runs beside the decoder's backward (sparse.set_wgrad_overlap(2)).  OFF by default.
struct AttnFork {
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
static AttnFork* attn_fork_of_device() {
    static AttnFork forks[64];
    static std::mutex mu;
    static const bool on = [] { const char* e = getenv("U3D_ATTN_FORK"); return e && atoi(e) != 0; }();
    if (!on) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    AttnFork& f = forks[dev & 63];
    std::lock_guard<std::mutex> lk(mu);
    if (!f.side) {
        if (hipStreamCreateWithFlags(&f.side, hipStreamNonBlocking) != hipSuccess) { f.side = nullptr; return nullptr; }
        hipEventCreateWithFlags(&f.fork, hipEventDisableTiming);
        hipEventCreateWithFlags(&f.join, hipEventDisableTiming);
    }
    return &f;
}

void attn_bwd_x3_launch(const float* qkv, const float* out, const float* dout, const float* lse, const int32_t* cu, int B, int max_len,
                        int64_t n_total, int H, float scale, float* dqkv, float* delta_ws, hipStream_t s, int planes) {
    if (planes == 16) hipLaunchKernelGGL(attn_delta_x16_k, dim3((unsigned)ceil_div(n_total * H, 256)), dim3(256), 0, s, (const __bf16*)out, (const __bf16*)dout, n_total, H, delta_ws);
    else hipLaunchKernelGGL(attn_delta_x3_k, dim3((unsigned)ceil_div(n_total * H, 256)), dim3(256), 0, s, out, dout, n_total, H, delta_ws);
    const int n_tiles = (max_len + 63) / 64;
    const dim3 grid((unsigned)(((H * B + 7) / 8) * 8 * n_tiles));
    AttnFork* f = attn_fork_of_device();
    hipStream_t sq = s;
    if (f) {
        hipEventRecord(f->fork, s);
        hipStreamWaitEvent(f->side, f->fork, 0);
        sq = f->side;
    }
    if (planes == 16) {
        hipLaunchKernelGGL((attn_bwd_dq_x3_k<1, true>), grid, dim3(256), 0, sq, (const __bf16*)qkv, (const __bf16*)dout, lse, (const float*)delta_ws, cu, H, scale, (__bf16*)dqkv, n_total, B, n_tiles);
        hipLaunchKernelGGL((attn_bwd_dkv_x3_k<1, true>), grid, dim3(256), 0, s, (const __bf16*)qkv, (const __bf16*)dout, lse, (const float*)delta_ws, cu, H, scale, (__bf16*)dqkv, n_total, B, n_tiles);
    } else if (planes == 1) {
        hipLaunchKernelGGL((attn_bwd_dq_x3_k<1>), grid, dim3(256), 0, sq, qkv, dout, lse, (const float*)delta_ws, cu, H, scale, dqkv, n_total, B, n_tiles);
        hipLaunchKernelGGL((attn_bwd_dkv_x3_k<1>), grid, dim3(256), 0, s, qkv, dout, lse, (const float*)delta_ws, cu, H, scale, dqkv, n_total, B, n_tiles);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_x3_k<3>), grid, dim3(256), 0, sq, qkv, dout, lse, (const float*)delta_ws, cu, H, scale, dqkv, n_total, B, n_tiles);
        hipLaunchKernelGGL((attn_bwd_dkv_x3_k<3>), grid, dim3(256), 0, s, qkv, dout, lse, (const float*)delta_ws, cu, H, scale, dqkv, n_total, B, n_tiles);
    }
    if (f) {
        hipEventRecord(f->join, f->side);
        hipStreamWaitEvent(s, f->join, 0);
    }
}

}  // namespace u3d

using namespace u3d;

extern "C" {

// bf16-operand form (BASELINE configs[2]; the reference's `--amp` run through nn.MultiheadAttention, tools/train.py:86-99): the
// same kernels with ONE plane per operand, rounded to nearest even where the three-plane form splits; qkv / out / gradients stay
// fp32 in HBM, accumulation and softmax stay fp32.
int u3d_attn_varlen_fwd_bf16(const float* qkv, const int32_t* cu_seqlens, int B, int max_len, int64_t n_total, int H, int hd,
                             float scale, float* out, float* lse, double flops_hint, u3d_stream_t stream) {
    if (!qkv || !cu_seqlens || !out || !lse || B <= 0 || H <= 0 || n_total <= 0) return U3D_EINVAL;
    if (hd != 32) { set_error("attn: head_dim %d unsupported (32 only)", hd); return U3D_EUNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_ATTN_FWD, s, flops_hint);
    if (max_len <= 0) return U3D_OK;
    attn_fwd_x3_launch(qkv, cu_seqlens, B, max_len, n_total, H, scale, out, lse, s, 1);
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
    attn_bwd_x3_launch(qkv, out, dout, lse, cu_seqlens, B, max_len, n_total, H, scale, dqkv, delta_ws, s, 1);
    return check_launch("attn_bwd_bf16");
}

// bf16 TENSORS (include/u3d.h K14b): qkv / out / dout / dqkv are bf16 in HBM -- what the reference's autocast hands to and takes from
// nn.MultiheadAttention (tools/train.py:86-99, unidet3d/encoder.py:19-21); lse / delta, softmax and all accumulators fp32.
int u3d_attn_varlen_fwd_b16(const void* qkv, const int32_t* cu_seqlens, int B, int max_len, int64_t n_total, int H, int hd,
                            float scale, void* out, float* lse, double flops_hint, u3d_stream_t stream) {
    if (!qkv || !cu_seqlens || !out || !lse || B <= 0 || H <= 0 || n_total <= 0) return U3D_EINVAL;
    if (hd != 32) { set_error("attn: head_dim %d unsupported (32 only)", hd); return U3D_EUNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_ATTN_FWD, s, flops_hint);
    if (max_len <= 0) return U3D_OK;
    attn_fwd_x3_launch((const float*)qkv, cu_seqlens, B, max_len, n_total, H, scale, (float*)out, lse, s, 16);
    return check_launch("attn_fwd_b16");
}

int u3d_attn_varlen_bwd_b16(const void* qkv, const void* out, const void* dout, const float* lse, const int32_t* cu_seqlens,
                            int B, int max_len, int64_t n_total, int H, int hd, float scale, void* dqkv, float* delta_ws,
                            double flops_hint, u3d_stream_t stream) {
    if (!qkv || !out || !dout || !lse || !cu_seqlens || !dqkv || !delta_ws || B <= 0 || H <= 0 || n_total <= 0) return U3D_EINVAL;
    if (hd != 32) { set_error("attn: head_dim %d unsupported (32 only)", hd); return U3D_EUNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_ATTN_BWD, s, flops_hint);
    if (max_len <= 0) return U3D_OK;
    attn_bwd_x3_launch((const float*)qkv, (const float*)out, (const float*)dout, lse, cu_seqlens, B, max_len, n_total, H, scale, (float*)dqkv, delta_ws, s, 16);
    return check_launch("attn_bwd_b16");
}

}  // extern "C"
