// Dense fp32 GEMMs of the decoder (nn.Linear forward / input-gradient / weight-gradient,
// unidet3d/encoder.py:19-21,55-61,138-140,153-155,163) on v_mfma_f32_32x32x2_f32.
// The query decoder runs over ~16k packed rows with K, N in {32, 256, 768, 1024}: tall-skinny fp32
// problems.  Measured on MI355X (tools/prof_gemm.py, M = 16000): the NT kernel (128x128 or 128x64 LDS tiles,
// buffer-load staging, one barrier per 16-deep K-step) reaches 76-103 TF/s, the TN kernel 32-75 TF/s; hipBLASLt
// 82-121 / 24-95 TF/s in isolation.  The decoder and the 1x1 skip convolutions run on these kernels, which keeps the
// whole Linear path (forward, dX, dW + bias gradient) behind the C ABI.
//
//   gemm_nt:  C[M,N] = A[M,K] . W[N,K]^T (+ bias[N])            (forward; dX = dY . (W^T)^T with W^T from u3d_transpose)
//   gemm_tn:  C[N,K] = A[M,N]^T . B[M,K]                        (weight gradient; reduction over the M rows is
//                                                                split over workgroups, partials summed in a fixed order)
// K-dim permutation as in the sparse conv kernels: lane (i, h) of the 32x32x2 MFMA feeds step s with
// k = 8h + s, so its 8 A (or B) values of a 16-deep K-step are two contiguous float4 in LDS.
#include "u3d_common.h"

namespace u3d {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define U3D_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int GT = 128;        // macro tile (both dims)
constexpr int GK = 16;         // K-step
constexpr int GLD = GK + 4;    // padded LDS row of the NT tiles

// TN = 128 or 64 output columns per workgroup: the decoder's GEMMs are tall and skinny (M ~ 16k, N = 256..1024), and
// with 128x128 tiles N = 256 gives only 250 workgroups for 256 CUs (one wave per SIMD, nothing to hide the barrier
// behind); 128x64 tiles double the workgroup count.  Global -> LDS staging uses raw buffer loads from a descriptor based
// at the tile origin: the K offset of a step is the instruction's scalar offset, so the per-thread address is loop
// invariant (no VALU in the K loop for addressing -- VALU time adds to MFMA time on this hardware), and rows past M / N
// read as zeros through the descriptor's bounds check instead of a compare + select.  The epilogue stores through a
// bounds-checked descriptor the same way (row offset scalar, column offset per lane, out-of-range lanes dropped).
// Epilogues (EPI) -- the decoder's MLPs (unidet3d/encoder.py:55-61 FFN, :138-140 input_proj, :153-155 outs_cls) without
// separate elementwise kernels:
//   0  C = acc + bias                                   plain nn.Linear
//   1  C = relu(acc + bias)                             Linear + ReLU
//   2  pre = acc + bias, C = gelu(pre)                  Linear + GELU (erf form, torch's default); pre is kept for backward
//   3  C = aux > 0 ? acc : 0                            input gradient through ReLU   (aux = the ReLU output)
//   4  C = acc * gelu'(aux)                             input gradient through GELU   (aux = the pre-activation)
//   5  C = acc + aux                                    a second gradient contribution added in place of a separate add kernel
// erf GELU with ONE exponential per element: erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2), t = 1 / (1 + p z), z = |x| / sqrt 2
// (Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7 -- below fp32 resolution of the products it enters), and exp(-z^2) =
// exp(-x^2 / 2) is also the Gaussian of the derivative.  libm's erff costs ~3x the VALU instructions, and VALU time in a
// GEMM epilogue is not hidden (one wave per SIMD): measured +95 us on the [16k x 1024] hidden-gradient GEMM with erff.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
    const float ax = fabsf(x), e = __expf(-0.5f * x * x);
    const float t = __frcp_rn(1.f + 0.3275911f * 0.70710678118654752440f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float tail = 0.5f * poly * e;                     // 0.5 * erfc(|x| / sqrt 2): Phi(-|x|) without cancellation
    cdf = x >= 0.f ? 1.f - tail : tail;
    pdf = 0.39894228040143267794f * e;
}
__device__ __forceinline__ float gelu_f(float x) { float c, p; gelu_parts(x, c, p); return x * c; }
__device__ __forceinline__ float gelu_grad_f(float x) { float c, p; gelu_parts(x, c, p); return c + x * p; }

// Shared epilogue of the NT kernels.  D layout of 32x32 MFMAs (dtype independent): col = lane & 31,
// row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
template <int TN, int EPI, int TM = GT>
__device__ __forceinline__ void nt_epilogue(const f32x16 (&acc)[TM / 64][TN / 64], float* __restrict__ C, const float* __restrict__ bias,
                                            const float* __restrict__ aux, float* __restrict__ pre, int64_t m0, int n0, int rows_a, int N,
                                            int wr, int wc, int i32, int kh) {
    constexpr int NB = TN / 64, TA = TM / 64;
    const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(C + m0 * N, (int64_t)rows_a * N * 4);
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc((EPI == 2 ? (const float*)pre : aux) + (EPI >= 2 ? m0 * N : 0), (int64_t)rows_a * N * 4);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int n = n0 + wc * (TN / 2) + b * 32 + i32;
        const float bv = (bias && n < N) ? bias[n] : 0.f;
        const int vc = n < N ? (4 * kh * N + n) * 4 : 0x7fffffff;        // columns past N: dropped by the bounds check
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * (TM / 2) + a * 32 + (r & 3) + 8 * (r >> 2);
                float v = acc[a][b][r] + bv;
                if constexpr (EPI == 1) v = fmaxf(v, 0.f);
                if constexpr (EPI == 2) {
                    float h = v;
                    asm volatile("" : "+v"(h));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h), rs_x, vc, row * N * 4, 0);
                    v = gelu_f(v);
                }
                if constexpr (EPI == 3 || EPI == 4 || EPI == 5) {       // rows / columns outside the tile read as 0 through the descriptor
                    const float x = __builtin_bit_cast(float, bload32(rs_x, vc, row * N * 4));
                    v = EPI == 3 ? (x > 0.f ? v : 0.f) : (EPI == 4 ? v * gelu_grad_f(x) : v + x);
                }
                asm volatile("" : "+v"(v));          // see gemm_tn_k: keeps the store builtin from mis-selecting the vector element
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_c, vc, row * N * 4, 0);
            }
    }
}

// TM x TN macro tile (TM = 128 or 64 rows: wave (wr, wc) owns TM/2 x TN/2), K-step GK
template <int TN, int EPI, int TM = GT>
__global__ __launch_bounds__(256) void gemm_nt_k(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                                                 float* __restrict__ C, int64_t M, int N, int K, const float* __restrict__ aux,
                                                 float* __restrict__ pre) {
    constexpr int NB = TN / 64, TA = TM / 64;    // 32-column / 32-row MFMA tiles per wave
    __shared__ __attribute__((aligned(16))) float As[2][TM * GLD];
    __shared__ __attribute__((aligned(16))) float Bs[2][TN * GLD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, i32 = lane & 31, kh = lane >> 5;
    // 1-D grid, XCD-aware: workgroup b runs on XCD b % 8 (own 4 MB L2); XCD x takes a contiguous chunk of tile ids with the
    // column tiles of one row tile next to each other, so the A row tile (GT x K, read by every column tile: PMC showed 196 MB
    // fetched per launch for 76 MB of operands with the 2-D grid) comes from HBM once and from that XCD's L2 afterwards
    const int nt_ = (N + TN - 1) / TN;
    const int64_t wid_ = xcd_swizzle(blockIdx.x, gridDim.x);
    const int64_t m0 = (wid_ / nt_) * TM;
    const int n0 = (int)(wid_ % nt_) * TN;
    const int rows_a = (int)min((int64_t)TM, M - m0), rows_b = min(TN, N - n0);
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(A + m0 * K, (int64_t)rows_a * K * 4);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(W + (int64_t)n0 * K, (int64_t)rows_b * K * 4);
    // staging map: thread -> (row = tid>>2 (+64), float4 column = tid&3)
    const int srow = tid >> 2, sc4 = tid & 3;
    const int vo = (srow * K + sc4 * 4) * 4, vstep = 64 * K * 4;
    f32x4 ra[TA], rb[NB];
    auto gload = [&](int kt) {
#pragma unroll
        for (int j = 0; j < TA; ++j) ra[j] = bload128(rs_a, vo + j * vstep, kt * (GK * 4));
#pragma unroll
        for (int j = 0; j < NB; ++j) rb[j] = bload128(rs_b, vo + j * vstep, kt * (GK * 4));
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < TA; ++j) *reinterpret_cast<f32x4*>(&As[buf][(srow + 64 * j) * GLD + sc4 * 4]) = ra[j];
#pragma unroll
        for (int j = 0; j < NB; ++j) *reinterpret_cast<f32x4*>(&Bs[buf][(srow + 64 * j) * GLD + sc4 * 4]) = rb[j];
    };
    f32x16 acc[TA][NB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int nk = K / GK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        f32x4 af[TA][2], bf[NB][2];
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            const float* ap = &As[buf][(wr * (TM / 2) + t * 32 + i32) * GLD + kh * 8];
            af[t][0] = *reinterpret_cast<const f32x4*>(ap);
            af[t][1] = *reinterpret_cast<const f32x4*>(ap + 4);
        }
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const float* bp = &Bs[buf][(wc * (TN / 2) + t * 32 + i32) * GLD + kh * 8];
            bf[t][0] = *reinterpret_cast<const f32x4*>(bp);
            bf[t][1] = *reinterpret_cast<const f32x4*>(bp + 4);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int a = 0; a < TA; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[a][b] = U3D_MFMA32(af[a][h][c], bf[b][h][c], acc[a][b]);
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    nt_epilogue<TN, EPI, TM>(acc, C, bias, aux, pre, m0, n0, rows_a, N, wr, wc, i32, kh);
}

// bf16-operand form of gemm_nt_k (BASELINE configs[2]; the reference's `--amp` Linear layers): A and W are fp32 in HBM, rounded
// to bf16 (RNE) while they are staged into LDS, multiplied by v_mfma_f32_32x32x16_bf16 with fp32 accumulation; same tiling,
// descriptors and epilogues.  A K-step is 32 deep (two MFMAs per tile pair); lane (i, h) holds k = 8h .. 8h+7 of each 16-deep
// half, one 16-byte LDS read.  80-byte LDS rows keep the 16-byte reads of 32 consecutive rows conflict-free.
constexpr int GKH = 32;        // K-step of the bf16 kernel
constexpr int GLH = GKH + 8;   // padded LDS row (halves)

template <int TN, int EPI, int TM = GT>
__global__ __launch_bounds__(256) void gemm_nt_bf16_k(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                                                      float* __restrict__ C, int64_t M, int N, int K, const float* __restrict__ aux,
                                                      float* __restrict__ pre) {
    constexpr int NB = TN / 64, TA = TM / 64;
    __shared__ __attribute__((aligned(16))) __bf16 As[2][TM * GLH];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][TN * GLH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, i32 = lane & 31, kh = lane >> 5;
    // 1-D grid, XCD-aware: workgroup b runs on XCD b % 8 (own 4 MB L2); XCD x takes a contiguous chunk of tile ids with the
    // column tiles of one row tile next to each other, so the A row tile (GT x K, read by every column tile: PMC showed 196 MB
    // fetched per launch for 76 MB of operands with the 2-D grid) comes from HBM once and from that XCD's L2 afterwards
    const int nt_ = (N + TN - 1) / TN;
    const int64_t wid_ = xcd_swizzle(blockIdx.x, gridDim.x);
    const int64_t m0 = (wid_ / nt_) * TM;
    const int n0 = (int)(wid_ % nt_) * TN;
    const int rows_a = (int)min((int64_t)TM, M - m0), rows_b = min(TN, N - n0);
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(A + m0 * K, (int64_t)rows_a * K * 4);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(W + (int64_t)n0 * K, (int64_t)rows_b * K * 4);
    // staging map: thread -> (row = tid>>2 (+64), 8 floats at column 8 * (tid&3))
    const int srow = tid >> 2, sc8 = tid & 3;
    const int vo = (srow * K + sc8 * 8) * 4, vstep = 64 * K * 4;
    f32x4 ra[TA][2], rb[NB][2];
    auto gload = [&](int kt) {
#pragma unroll
        for (int j = 0; j < TA; ++j) {
            ra[j][0] = bload128(rs_a, vo + j * vstep, kt * (GKH * 4));
            ra[j][1] = bload128(rs_a, vo + j * vstep + 16, kt * (GKH * 4));
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            rb[j][0] = bload128(rs_b, vo + j * vstep, kt * (GKH * 4));
            rb[j][1] = bload128(rs_b, vo + j * vstep + 16, kt * (GKH * 4));
        }
    };
    auto cvt8 = [](const f32x4& lo, const f32x4& hi) {
        return bf16x8{(__bf16)lo[0], (__bf16)lo[1], (__bf16)lo[2], (__bf16)lo[3], (__bf16)hi[0], (__bf16)hi[1], (__bf16)hi[2], (__bf16)hi[3]};
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < TA; ++j) *reinterpret_cast<bf16x8*>(&As[buf][(srow + 64 * j) * GLH + sc8 * 8]) = cvt8(ra[j][0], ra[j][1]);
#pragma unroll
        for (int j = 0; j < NB; ++j) *reinterpret_cast<bf16x8*>(&Bs[buf][(srow + 64 * j) * GLH + sc8 * 8]) = cvt8(rb[j][0], rb[j][1]);
    };
    f32x16 acc[TA][NB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int nk = K / GKH;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 af[TA], bf[NB];
#pragma unroll
            for (int t = 0; t < TA; ++t) af[t] = *reinterpret_cast<const bf16x8*>(&As[buf][(wr * (TM / 2) + t * 32 + i32) * GLH + h * 16 + kh * 8]);
#pragma unroll
            for (int t = 0; t < NB; ++t) bf[t] = *reinterpret_cast<const bf16x8*>(&Bs[buf][(wc * (TN / 2) + t * 32 + i32) * GLH + h * 16 + kh * 8]);
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    nt_epilogue<TN, EPI, TM>(acc, C, bias, aux, pre, m0, n0, rows_a, N, wr, wc, i32, kh);
}

// gemm_nt_k with the fp32 products on the bf16 pipe (u3d_common.h, "bf16x3"): the staged fp32 rows are split into three exact
// bf16 planes, a tile pair takes the six MFMAs h.h, h.m, m.h, h.l, m.m, l.h (smallest terms first) -- 6 x 32 cycles for a
// 32 x 32 x 16 block instead of the 8 x 64 cycles of v_mfma_f32_32x32x2_f32.  With the matrix time cut to 3/8 the kernel must
// not be latency bound, so the loop is built differently from gemm_nt_k:
//   * ONE LDS stage (3 planes x (TM + TN) rows x 80 B = 60 KB at 128 x 128, 45 KB at 128 x 64) instead of two, so two -- at
//     128 x 64, with the register bound below, three -- workgroups share a CU and one computes while another stages (their
//     barriers are independent);
//   * the raw fp32 rows of stage kt+1 wait in registers while stage kt is multiplied, are split AFTER the MFMAs were issued
//     (VALU work under the matrix pipe, this wave's own and the partner workgroup's), and the loads of stage kt+2 are issued
//     before the barrier: every global load has a full stage of matrix work to land.
#ifndef U3D_NTX_ABL
#define U3D_NTX_ABL 0          // timing ablations (wrong results): 1 no MFMAs, 2 no split arithmetic, 4 no global loads in the loop, 8 no LDS stores in the loop
#endif
// WP: the W operand arrives ALREADY split -- `Wpl` = three bf16 planes [3][N][K] written once per training step for all weights
// (planes_batch_k below: the same split3_x8 on the same 8-element groups, so the staged planes and the results are bit-identical) --
// and the thread that stages a W row piece loads three 16-byte plane pieces instead of 32 bytes of fp32 and runs no split for it: a
// third of the kernel's split arithmetic (13 VALU per pair of values, issued in the matrix pipe's time) at TM = 128, TN = 64.
template <int TN, int EPI, int TM = GT, bool WP = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TM == 128 ? 3 : 1))) void gemm_nt_x3_k(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                                                    float* __restrict__ C, int64_t M, int N, int K, const float* __restrict__ aux,
                                                    float* __restrict__ pre, const void* __restrict__ Wpl) {
    constexpr int NB = TN / 64, TA = TM / 64;
    __shared__ __attribute__((aligned(16))) __bf16 As[3][TM * GLH];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[3][TN * GLH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, i32 = lane & 31, kh = lane >> 5;
    const int nt_ = (N + TN - 1) / TN;                       // XCD-aware 1-D grid, see gemm_nt_k
    const int64_t wid_ = xcd_swizzle(blockIdx.x, gridDim.x);
    const int64_t m0 = (wid_ / nt_) * TM;
    const int n0 = (int)(wid_ % nt_) * TN;
    const int rows_a = (int)min((int64_t)TM, M - m0), rows_b = min(TN, N - n0);
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(A + m0 * K, (int64_t)rows_a * K * 4);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(W + (int64_t)n0 * K, (int64_t)rows_b * K * 4);
    // staging map: thread -> (row = tid>>2 (+64 j), 8 floats at column 8 * (tid&3))
    const int srow = tid >> 2, sc8 = tid & 3;
    const int vo = (srow * K + sc8 * 8) * 4, vstep = 64 * K * 4;
    // pre-split W: plane q of rows n0 .. n0 + rows_b (loads past the last row return zeros, like the fp32 rows)
    const char* wpl = reinterpret_cast<const char*>(Wpl);
    const __amdgpu_buffer_rsrc_t rs_p0 = make_rsrc(wpl + ((int64_t)0 * N + n0) * K * 2, (int64_t)rows_b * K * 2);
    const __amdgpu_buffer_rsrc_t rs_p1 = make_rsrc(wpl + ((int64_t)1 * N + n0) * K * 2, (int64_t)rows_b * K * 2);
    const __amdgpu_buffer_rsrc_t rs_p2 = make_rsrc(wpl + ((int64_t)2 * N + n0) * K * 2, (int64_t)rows_b * K * 2);
    const int vop = (srow * K + sc8 * 8) * 2, vstep_p = 64 * K * 2;
    f32x4 ra[TA][2], rb[NB][WP ? 3 : 2];
    auto gload = [&](int kt) {
#pragma unroll
        for (int j = 0; j < TA; ++j) {
            ra[j][0] = bload128(rs_a, vo + j * vstep, kt * (GKH * 4));
            ra[j][1] = bload128(rs_a, vo + j * vstep + 16, kt * (GKH * 4));
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if constexpr (WP) {
                rb[j][0] = bload128(rs_p0, vop + j * vstep_p, kt * (GKH * 2));
                rb[j][1] = bload128(rs_p1, vop + j * vstep_p, kt * (GKH * 2));
                rb[j][2] = bload128(rs_p2, vop + j * vstep_p, kt * (GKH * 2));
            } else {
                rb[j][0] = bload128(rs_b, vo + j * vstep, kt * (GKH * 4));
                rb[j][1] = bload128(rs_b, vo + j * vstep + 16, kt * (GKH * 4));
            }
        }
    };
    bf16x8 pa[TA][3], pb[NB][3];
    auto split = [&]() {
        if constexpr (U3D_NTX_ABL & 2) {
#pragma unroll
            for (int j = 0; j < TA; ++j) { pa[j][0] = __builtin_bit_cast(bf16x8, ra[j][0]); pa[j][1] = __builtin_bit_cast(bf16x8, ra[j][1]); pa[j][2] = pa[j][0]; }
#pragma unroll
            for (int j = 0; j < NB; ++j) { pb[j][0] = __builtin_bit_cast(bf16x8, rb[j][0]); pb[j][1] = __builtin_bit_cast(bf16x8, rb[j][1]); pb[j][2] = pb[j][0]; }
            return;
        }
#pragma unroll
        for (int j = 0; j < TA; ++j) split3_x8(ra[j][0], ra[j][1], pa[j]);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if constexpr (WP) {
#pragma unroll
                for (int q = 0; q < 3; ++q) pb[j][q] = __builtin_bit_cast(bf16x8, rb[j][q]);
            } else {
                split3_x8(rb[j][0], rb[j][1], pb[j]);
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int j = 0; j < TA; ++j) *reinterpret_cast<bf16x8*>(&As[q][(srow + 64 * j) * GLH + sc8 * 8]) = pa[j][q];
#pragma unroll
            for (int j = 0; j < NB; ++j) *reinterpret_cast<bf16x8*>(&Bs[q][(srow + 64 * j) * GLH + sc8 * 8]) = pb[j][q];
        }
    };
    f32x16 acc[TA][NB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // The five low-order plane products accumulate in their OWN running tile `lo`, added to `acc` once at the end: a bf16 MFMA
    // truncates its 32 products at the exponent of its C operand, always towards zero (tools/bias_probe.py: -1.2e-8 mean error
    // at K = 256 with everything in one tile -- a coherent bias that reductions over many rows downstream do not average out);
    // products 2^-8 .. 2^-16 below the running sum lose the most, against their own small tile they lose nothing that matters.
    f32x16 lo[TA][NB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) lo[a][b][r] = 0.f;
    const int nk = K / GKH;
#ifdef U3D_NTX_TRACE          // phase timestamps of wave 0 (tools/trace_gemm.py; `pre` is the trace buffer, EPI 0 only)
    uint64_t tr_[12];
#define U3D_TR(i) tr_[i] = __builtin_amdgcn_s_memtime()
#define U3D_TRK(i) if (kt == 3) U3D_TR(i)
#else
#define U3D_TR(i)
#define U3D_TRK(i)
#endif
    U3D_TR(0);
    gload(0);
    split();
    if (nk > 1) gload(1);
    lstore();
    __syncthreads();
    U3D_TR(1);
    for (int kt = 0; kt < nk; ++kt) {
        U3D_TRK(2);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 af[3][TA], bf[3][NB];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int t = 0; t < TA; ++t) af[q][t] = *reinterpret_cast<const bf16x8*>(&As[q][(wr * (TM / 2) + t * 32 + i32) * GLH + h * 16 + kh * 8]);
#pragma unroll
                for (int t = 0; t < NB; ++t) bf[q][t] = *reinterpret_cast<const bf16x8*>(&Bs[q][(wc * (TN / 2) + t * 32 + i32) * GLH + h * 16 + kh * 8]);
            }
#pragma unroll
            for (int o = 2; o >= 0; --o)           // plane-order sum qa + qb = o: smallest terms first
#pragma unroll
                for (int qa = 0; qa <= o; ++qa)
#pragma unroll
                    for (int a = 0; a < TA; ++a)
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            if constexpr (U3D_NTX_ABL & 1) { if (o == 2 && qa == 0) acc[a][b][0] += (float)af[0][a][0] + (float)af[1][a][1] + (float)af[2][a][2] + (float)bf[0][b][0] + (float)bf[1][b][1] + (float)bf[2][b][2]; }
                            else if (o == 0) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[qa][a], bf[o - qa][b], acc[a][b], 0, 0, 0);
                            else lo[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[qa][a], bf[o - qa][b], lo[a][b], 0, 0, 0);
                        }
        }
        if (kt + 1 < nk) split();                  // stage kt+1 (loaded one iteration ago)
        U3D_TRK(3);
        if (kt + 2 < nk && !(U3D_NTX_ABL & 4)) gload(kt + 2);
        U3D_TRK(4);
        __syncthreads();                           // every wave has read stage kt
        U3D_TRK(5);
        if (kt + 1 < nk && !(U3D_NTX_ABL & 8)) lstore();
        U3D_TRK(6);
        __syncthreads();
        U3D_TRK(7);
    }
    U3D_TR(8);
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] += lo[a][b];
    nt_epilogue<TN, EPI, TM>(acc, C, bias, aux, pre, m0, n0, rows_a, N, wr, wc, i32, kh);
#ifdef U3D_NTX_TRACE
    if constexpr (EPI == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        U3D_TR(9);
        if (pre && tid == 0) {
            unsigned hw;
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            hw |= (xcc & 0xf) << 28;
            uint64_t* o = reinterpret_cast<uint64_t*>(pre) + (int64_t)blockIdx.x * 12;
#pragma unroll
            for (int i = 0; i < 10; ++i) o[i] = tr_[i];
            o[10] = hw;
        }
    }
#endif
#undef U3D_TR
#undef U3D_TRK
}

// partial[s][n][k] = sum over this split's rows of A[m][n] * B[m][k].  T = 128 or 64 output rows AND columns per workgroup:
// the weight gradients of this decoder are small (256 x 256, 768 x 256, 256 x 32, 19 x 256 ...) -- with 128 x 128 tiles a
// 256 x 256 output has 4 tiles and needs ~96 row splits to fill 256 CUs, i.e. 25 MB of partials written and read back for 0.26 MB
// of result; 64 x 64 tiles quarter the splits and the partial traffic at the price of twice the LDS reads per MFMA.
template <int T>
__global__ __launch_bounds__(256) void gemm_tn_k(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ partial,
                                                 int colsum, int64_t M, int N, int K, int64_t rows_per_split, int S) {
    // workgroup -> (row split, output tile), XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs, each with its
    // own L2; every tile of one row split reads the same rows of A and B, so all of them are placed on ONE XCD (ids congruent
    // mod 8, adjacent in that XCD's order) and the operands come from HBM once instead of once per tile
    const int tiles_n = (N + T - 1) / T, tiles = tiles_n * ((K + T - 1) / T);
    const int slot = blockIdx.x >> 3, tile = slot % tiles;
    const int split = (slot / tiles) * 8 + (blockIdx.x & 7);
    if (split >= S) return;
    constexpr int LD = T + 4;                   // padded LDS row of the tiles ([16 rows][T cols])
    constexpr int NF = T / 64;                  // 32-wide MFMA tiles per wave and operand
    constexpr int TPR = T / 4;                  // staging threads per tile row (float4 each)
    constexpr int RPP = 256 / TPR;              // rows per staging pass
    constexpr int NP = GK / RPP;                // staging passes per K-step
    __shared__ __attribute__((aligned(16))) float As[2][GK * LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GK * LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, i32 = lane & 31, kh = lane >> 5;
    const int n0 = (tile % tiles_n) * T, k0 = (tile / tiles_n) * T;
    const int64_t mlo = (int64_t)split * rows_per_split;
    const int64_t mhi = min(M, mlo + rows_per_split);
    const int rows = (int)max((int64_t)0, mhi - mlo);
    // buffer loads from descriptors based at this split's first row: the row block of a step is the scalar offset, rows past
    // the split and columns past N / K read as zeros
    const int srow = tid / TPR, sc4 = tid % TPR;
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(A + mlo * N, (int64_t)rows * N * 4);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(B + mlo * K, (int64_t)rows * K * 4);
    const int va = n0 + sc4 * 4 < N ? (srow * N + n0 + sc4 * 4) * 4 : 0x7fffffff;
    const int vb = k0 + sc4 * 4 < K ? (srow * K + k0 + sc4 * 4) * 4 : 0x7fffffff;
    const bool ca = va != 0x7fffffff, cb = vb != 0x7fffffff;
    f32x4 ra[NP], rb[NP];
    auto gload = [&](int t) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            ra[j] = bload128(rs_a, ca ? va + j * RPP * N * 4 : va, t * (GK * N * 4));
            rb[j] = bload128(rs_b, cb ? vb + j * RPP * K * 4 : vb, t * (GK * K * 4));
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            *reinterpret_cast<f32x4*>(&As[buf][(srow + RPP * j) * LD + sc4 * 4]) = ra[j];
            *reinterpret_cast<f32x4*>(&Bs[buf][(srow + RPP * j) * LD + sc4 * 4]) = rb[j];
        }
    };
    f32x16 acc[NF][NF];
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
        for (int b = 0; b < NF; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int nt = (rows + GK - 1) / GK;
    const bool sums = colsum && k0 == 0 && tid < T;       // column sums of A (the bias gradient) ride along
    float csum = 0.f;
    if (nt > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) gload(t + 1);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int m = kh * 8 + s;                       // reduction index of this lane at step s
            float av[NF], bv[NF];
#pragma unroll
            for (int u = 0; u < NF; ++u) {
                av[u] = As[buf][m * LD + wr * (T / 2) + u * 32 + i32];
                bv[u] = Bs[buf][m * LD + wc * (T / 2) + u * 32 + i32];
            }
#pragma unroll
            for (int a = 0; a < NF; ++a)
#pragma unroll
                for (int b = 0; b < NF; ++b) acc[a][b] = U3D_MFMA32(av[a], bv[b], acc[a][b]);
        }
        if (sums) {
#pragma unroll
            for (int m = 0; m < GK; ++m) csum += As[buf][m * LD + tid];
        }
        if (t + 1 < nt) lstore(buf ^ 1);
        __syncthreads();
    }
    const int64_t pstride = (int64_t)N * K + (colsum ? N : 0);          // a split's block: [N*K] products, then [N] column sums
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(partial + (int64_t)split * pstride, (int64_t)N * K * 4);
#pragma unroll
    for (int b = 0; b < NF; ++b) {
        const int k = k0 + wc * (T / 2) + b * 32 + i32;
        const int vo = k < K ? ((n0 + 4 * kh) * K + k) * 4 : 0x7fffffff;        // rows past N fall off the end of the descriptor
#pragma unroll
        for (int a = 0; a < NF; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * (T / 2) + a * 32 + (r & 3) + 8 * (r >> 2);
                // the element goes through an opaque VGPR copy: handed a vector element directly, this compiler's buffer-store
                // builtin stores element 0 of the accumulator sixteen times (seen in the ISA, caught by the parity test)
                float v = acc[a][b][r];
                asm volatile("" : "+v"(v));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_o, vo, row * K * 4, 0);
            }
    }
    if (sums && n0 + tid < N) partial[(int64_t)split * pstride + (int64_t)N * K + n0 + tid] = csum;
}

// gemm_tn_k with fp32 products from three exact bf16 planes per operand (u3d_common.h "bf16x3"), round 4.  The reduction index is
// the ROW index of both operands, so a lane's eight k values are a COLUMN of the staged [32 rows][T cols] tiles.  What makes that
// affordable (round 3's form -- pair-packed planes read back with ds_read_b32, every wave splitting the fragments it read -- was not):
//   * every staged fp32 value is split ONCE, by the thread that loaded it, into three row-major bf16 plane tiles;
//   * column fragments are ds_read_b64_tr_b16 (a 16-lane group hands in a [4 rows][16 cols] block, lane t receives column t): two
//     reads per plane and fragment, k slot 8g + e <-> row 4g + (e & 3) + 16 (e >> 2) for both operands.  Rows are padded by 32 bytes
//     (odd multiples of 32 B): the eight rows a half-wave reads lie on eight different 32-byte bank groups;
//   * 128 x 64 (FFN weights) and 64 x 64 tiles: the low-order products keep accumulators of their own (the bf16 MFMA truncates at
//     its C operand's exponent, u3d_common.h) without leaving two waves per SIMD.
// One LDS stage, two barriers per 32-row trip (the raw rows of trip t+1 wait in registers during the products of trip t).
constexpr int TXK = 32;
template <int TA, int TB>          // output rows (columns of A) x output columns (columns of B) per workgroup
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TA == 128 ? 3 : 1))) void gemm_tn_x3_k(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ partial,
                                                    int colsum, int64_t M, int N, int K, int64_t rows_per_split, int S) {
    constexpr int LA = TA + 16, LB = TB + 16;              // halves per LDS row
    constexpr int PA = TXK * LA, PB = TXK * LB;            // halves per plane tile
    constexpr int TPA = TA / 4, TPB = TB / 4;              // staging threads per row (one float4 each)
    constexpr int RPA = 256 / TPA, RPB = 256 / TPB;        // rows per staging pass
    constexpr int NPA = TXK / RPA, NPB = TXK / RPB;        // passes per trip
    constexpr int NA = TA / 32, NB = TB / 32;              // 16-wide blocks per wave and operand (2 x 2 waves)
    __shared__ __attribute__((aligned(16))) __bf16 As[3 * PA];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[3 * PB];
    __shared__ float csum_s[RPA][TA];
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int tiles_n = (N + TA - 1) / TA, tiles = tiles_n * ((K + TB - 1) / TB);
    const int slot = blockIdx.x >> 3, tile = slot % tiles;                     // XCD placement as in gemm_tn_k
    const int split = (slot / tiles) * 8 + (blockIdx.x & 7);
    if (split >= S) return;
    const int tid = threadIdx.x, lane = tid & 63, t16 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int n0 = (tile % tiles_n) * TA, k0 = (tile / tiles_n) * TB;
    const int64_t mlo = (int64_t)split * rows_per_split;
    const int64_t mhi = min(M, mlo + rows_per_split);
    const int rows = (int)max((int64_t)0, mhi - mlo);
    const int rowa = tid / TPA, ca4 = tid % TPA, rowb = tid / TPB, cb4 = tid % TPB;
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(A + mlo * N, (int64_t)rows * N * 4);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(B + mlo * K, (int64_t)rows * K * 4);
    const int va = n0 + ca4 * 4 < N ? (rowa * N + n0 + ca4 * 4) * 4 : 0x7fffffff;        // rows past the split / columns past N, K read as zeros
    const int vb = k0 + cb4 * 4 < K ? (rowb * K + k0 + cb4 * 4) * 4 : 0x7fffffff;
    const bool oka = va != 0x7fffffff, okb = vb != 0x7fffffff;
    f32x4 ra[NPA], rb[NPB];
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    auto gload = [&](int t) {
#pragma unroll
        for (int j = 0; j < NPA; ++j) ra[j] = bload128(rs_a, oka ? va + j * RPA * N * 4 : va, t * (TXK * N * 4));
#pragma unroll
        for (int j = 0; j < NPB; ++j) rb[j] = bload128(rs_b, okb ? vb + j * RPB * K * 4 : vb, t * (TXK * K * 4));
    };
    auto put = [&](__bf16* tile, int plane, int off, const f32x4& v) {
        unsigned w0[3], w1[3];
        split3_pair(v[0], v[1], w0[0], w0[1], w0[2]);
        split3_pair(v[2], v[3], w1[0], w1[1], w1[2]);
#pragma unroll
        for (int q = 0; q < 3; ++q) *reinterpret_cast<uint2*>(tile + q * plane + off) = make_uint2(w0[q], w1[q]);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
            cs += ra[j];
            put(As, PA, (rowa + j * RPA) * LA + ca4 * 4, ra[j]);
        }
#pragma unroll
        for (int j = 0; j < NPB; ++j) put(Bs, PB, (rowb + j * RPB) * LB + cb4 * 4, rb[j]);
    };
    // column blk * 16 + t16 of a plane tile over rows {4g .. 4g+3} and {16+4g .. 16+4g+3}
    auto frag = [&](const __bf16* tile, int L, int blk) -> bf16x8 {
        const __bf16* s0 = tile + (4 * g + (t16 >> 2)) * L + blk * 16 + 4 * (t16 & 3);
        const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)s0);
        const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(s0 + 16 * L));
        return __builtin_bit_cast(bf16x8, s16x8{r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]});
    };
    f32x4 acc[NA][NB], lo[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) { acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[a][b] = acc[a][b]; }
    const int nt = (rows + TXK - 1) / TXK;
    if (nt > 0) {
        gload(0);
        lstore();
        if (nt > 1) gload(1);
    }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        bf16x8 fb[NB][3];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 3; ++q) fb[b][q] = frag(Bs + q * PB, LB, wc * NB + b);
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            bf16x8 fa[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) fa[q] = frag(As + q * PA, LA, wr * NA + a);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int o = 2; o >= 1; --o)               // plane-order sum qa + qb = o, smallest terms first
#pragma unroll
                    for (int qa = 0; qa <= o; ++qa)
                        lo[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[qa], fb[b][o - qa], lo[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[b][0], acc[a][b], 0, 0, 0);
            }
        }
        __syncthreads();                                   // every wave has read trip t
        if (t + 1 < nt) {
            lstore();                                      // trip t+1 (loaded one trip ago)
            if (t + 2 < nt) gload(t + 2);
        }
        __syncthreads();
    }
    // accumulator element r of lane (t16, g) in block (a, b): output row n0 + 16 (wr NA + a) + 4g + r, column k0 + 16 (wc NB + b) + t16
    const int64_t pstride = (int64_t)N * K + (colsum ? N : 0);
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(partial + (int64_t)split * pstride, (int64_t)N * K * 4);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int k = k0 + 16 * (wc * NB + b) + t16;
        const int vo = k < K ? ((n0 + 4 * g) * K + k) * 4 : 0x7fffffff;          // rows past N fall off the end of the descriptor
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[a][b][r] + lo[a][b][r];
                asm volatile("" : "+v"(v));                 // see gemm_tn_k
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_o, vo, (16 * (wr * NA + a) + r) * K * 4, 0);
            }
    }
    if (colsum && k0 == 0) {                               // column sums of A from the fp32 values: RPA row slots per column
#pragma unroll
        for (int c = 0; c < 4; ++c) csum_s[rowa][ca4 * 4 + c] = cs[c];
        __syncthreads();
        if (tid < TA && n0 + tid < N) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < RPA; ++r) v += csum_s[r][tid];
            partial[(int64_t)split * pstride + (int64_t)N * K + n0 + tid] = v;
        }
    }
}

// bf16-operand form of gemm_tn_k (weight gradients of the Linear layers under BASELINE configs[2]): A (= dY) and B (= X) are
// rounded to bf16 while they are staged, the reduction over the M rows runs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
// The reduction index is the ROW index of both operands, so a lane's 8 consecutive k values are a COLUMN of the staged
// [32 rows][128 cols] tile: two ds_read_b64_tr_b16 per fragment (a 16-lane group hands in a [4 rows][16 cols] block, lane t
// receives column t; rounds 2-3 used eight 2-byte reads and four packing instructions).  320-byte rows put the four rows a
// half-wave reads on four different 64-byte bank groups -- conflict free.  The column sums of A (bias gradient) are accumulated
// from the fp32 values before they are rounded.
constexpr int TKH = 32;              // rows per stage
constexpr int TLH = GT + 32;         // padded LDS row (halves)

__global__ __launch_bounds__(256) void gemm_tn_bf16_k(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ partial,
                                                      int colsum, int64_t M, int N, int K, int64_t rows_per_split) {
    __shared__ __attribute__((aligned(16))) __bf16 As[2][TKH * TLH];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][TKH * TLH];
    __shared__ float csum_s[8][GT];
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, i32 = lane & 31, kh = lane >> 5;
    const int n0 = blockIdx.x * GT, k0 = blockIdx.y * GT;
    const int64_t mlo = (int64_t)blockIdx.z * rows_per_split;
    const int64_t mhi = min(M, mlo + rows_per_split);
    const int rows = (int)max((int64_t)0, mhi - mlo);
    // staging map: thread -> (row = tid>>5 (+8 j, j < 4), float4 column = tid&31)
    const int srow = tid >> 5, sc4 = tid & 31;
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(A + mlo * N, (int64_t)rows * N * 4);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(B + mlo * K, (int64_t)rows * K * 4);
    const int va = n0 + sc4 * 4 < N ? (srow * N + n0 + sc4 * 4) * 4 : 0x7fffffff;
    const int vb = k0 + sc4 * 4 < K ? (srow * K + k0 + sc4 * 4) * 4 : 0x7fffffff;
    const bool ca = va != 0x7fffffff, cb = vb != 0x7fffffff;
    f32x4 ra[4], rb[4];
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    auto gload = [&](int t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j] = bload128(rs_a, ca ? va + j * 8 * N * 4 : va, t * (TKH * N * 4));
            rb[j] = bload128(rs_b, cb ? vb + j * 8 * K * 4 : vb, t * (TKH * K * 4));
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cs += ra[j];
            *reinterpret_cast<bf16x4*>(&As[buf][(srow + 8 * j) * TLH + sc4 * 4]) = bf16x4{(__bf16)ra[j][0], (__bf16)ra[j][1], (__bf16)ra[j][2], (__bf16)ra[j][3]};
            *reinterpret_cast<bf16x4*>(&Bs[buf][(srow + 8 * j) * TLH + sc4 * 4]) = bf16x4{(__bf16)rb[j][0], (__bf16)rb[j][1], (__bf16)rb[j][2], (__bf16)rb[j][3]};
        }
    };
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int troff = ((lane & 15) >> 2) * TLH + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);      // this lane's 8 bytes of its group's [4 rows][16 cols] block
    auto colfrag = [&](const __bf16* t, int row0, int col32) {       // column col32 + i32 over rows row0 .. row0 + 7
        const __bf16* p = t + row0 * TLH + col32 + troff;
        const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
        const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * TLH));
        return __builtin_bit_cast(bf16x8, s16x8{r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]});
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int nt = (rows + TKH - 1) / TKH;
    if (nt > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) gload(t + 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r0 = h * 16 + kh * 8;                 // this lane's 8 reduction rows of the 16-deep block
            bf16x8 af[2], bf[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                af[u] = colfrag(As[buf], r0, wr * 64 + u * 32);
                bf[u] = colfrag(Bs[buf], r0, wc * 64 + u * 32);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (t + 1 < nt) lstore(buf ^ 1);
        __syncthreads();
    }
    const int64_t pstride = (int64_t)N * K + (colsum ? N : 0);          // a split's block: [N*K] products, then [N] column sums
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(partial + (int64_t)blockIdx.z * pstride, (int64_t)N * K * 4);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int k = k0 + wc * 64 + b * 32 + i32;
        const int vo = k < K ? ((n0 + 4 * kh) * K + k) * 4 : 0x7fffffff;        // rows past N fall off the end of the descriptor
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2);
                float v = acc[a][b][r];
                asm volatile("" : "+v"(v));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_o, vo, row * K * 4, 0);
            }
    }
    if (colsum && blockIdx.y == 0) {       // column sums of A: 8 row-slots per column -> one value per column
#pragma unroll
        for (int c = 0; c < 4; ++c) csum_s[srow][sc4 * 4 + c] = cs[c];
        __syncthreads();
        if (tid < GT && n0 + tid < N) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) v += csum_s[r][tid];
            partial[(int64_t)blockIdx.z * pstride + (int64_t)N * K + n0 + tid] = v;
        }
    }
}

// sums the splits in a fixed order; float4 i < n4_main goes to C, the rest (the column sums) to C2
__global__ __launch_bounds__(256) void gemm_tn_reduce_k(const float* __restrict__ partial, int S, int64_t n4, int64_t n4_main, float* __restrict__ C,
                                                        float* __restrict__ C2) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        // four independent partial sums (loads in flight), combined in a fixed order
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        int s = 0;
        for (; s + 4 <= S; s += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 t = reinterpret_cast<const float4*>(partial)[(int64_t)(s + u) * n4 + i];
                v[u].x += t.x; v[u].y += t.y; v[u].z += t.z; v[u].w += t.w;
            }
        }
        for (; s < S; ++s) {
            const float4 t = reinterpret_cast<const float4*>(partial)[(int64_t)s * n4 + i];
            v[0].x += t.x; v[0].y += t.y; v[0].z += t.z; v[0].w += t.w;
        }
        float4* dst = i < n4_main ? reinterpret_cast<float4*>(C) + i : reinterpret_cast<float4*>(C2) + (i - n4_main);
        *dst = make_float4((v[0].x + v[1].x) + (v[2].x + v[3].x), (v[0].y + v[1].y) + (v[2].y + v[3].y),
                                                      (v[0].z + v[1].z) + (v[2].z + v[3].z), (v[0].w + v[1].w) + (v[2].w + v[3].w));
    }
}

// stand-alone GELU (erf) passes over [n] floats -- the unfused alternative to epilogues 2 / 4 (HBM-bound, full occupancy)
__global__ __launch_bounds__(256) void gelu_fwd_k(const float4* __restrict__ h, float4* __restrict__ a, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = h[i];
        a[i] = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
    }
}
__global__ __launch_bounds__(256) void gelu_bwd_k(const float4* __restrict__ da, const float4* __restrict__ h, float4* __restrict__ dh, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 g = da[i], v = h[i];
        dh[i] = make_float4(g.x * gelu_grad_f(v.x), g.y * gelu_grad_f(v.y), g.z * gelu_grad_f(v.z), g.w * gelu_grad_f(v.w));
    }
}

// out[c][r] = in[r][c]
__global__ __launch_bounds__(256) void transpose_k(const float* __restrict__ in, float* __restrict__ out, int R, int Ccols) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (by + j < R && bx + tx < Ccols) tile[j][tx] = in[(int64_t)(by + j) * Ccols + bx + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (bx + j < Ccols && by + tx < R) out[(int64_t)(bx + j) * R + by + tx] = tile[tx][j];
}

// Every transposed weight copy a step's input-gradient GEMMs need (dX = dY . W as an NT product over W^T), in ONE launch: desc[i] =
// {src [R][C], dst [C][R], R, C, first block}; block b belongs to the last descriptor whose first block is <= b (the pattern of
// u3d_weight_pack_batch).  Replaces ~31 transpose_k launches of ~4 us per training step.
struct TrDesc { const float* src; float* dst; int64_t R, C, block0; };
__global__ __launch_bounds__(256) void transpose_batch_k(const TrDesc* __restrict__ desc, int n_desc) {
    __shared__ float tile[32][33];
    int lo = 0, hi = n_desc;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (desc[mid].block0 <= (int64_t)blockIdx.x) lo = mid; else hi = mid;
    }
    const TrDesc d = desc[lo];
    const int R = (int)d.R, Ccols = (int)d.C;
    const int tiles_x = (Ccols + 31) / 32;
    const int rem = (int)((int64_t)blockIdx.x - d.block0);
    const int bx = (rem % tiles_x) * 32, by = (rem / tiles_x) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (by + j < R && bx + tx < Ccols) tile[j][tx] = d.src[(int64_t)(by + j) * Ccols + bx + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (bx + j < Ccols && by + tx < R) d.dst[(int64_t)(bx + j) * R + by + tx] = tile[tx][j];
}

// The three bf16 planes of every weight (and of its transposed copy) a step's NT products take as their W operand, in ONE launch:
// desc[i] = {src fp32 [n8 * 8], dst planes [3][n8 * 8] bf16, n8, first block}; thread = one group of 8 consecutive values, split
// by the SAME split3_x8 the GEMM kernels run on the fly, so the pre-split operand is bit-identical to the in-kernel one.
struct PlDesc { const float* src; bf16x8* dst; int64_t n8, block0; };
__global__ __launch_bounds__(256) void planes_batch_k(const PlDesc* __restrict__ desc, int n_desc) {
    int lo = 0, hi = n_desc;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (desc[mid].block0 <= (int64_t)blockIdx.x) lo = mid; else hi = mid;
    }
    const PlDesc d = desc[lo];
    const int64_t idx = ((int64_t)blockIdx.x - d.block0) * 256 + threadIdx.x;
    if (idx >= d.n8) return;
    const f32x4 a = *reinterpret_cast<const f32x4*>(d.src + idx * 8), b = *reinterpret_cast<const f32x4*>(d.src + idx * 8 + 4);
    bf16x8 out[3];
    split3_x8(a, b, out);
#pragma unroll
    for (int q = 0; q < 3; ++q) d.dst[q * d.n8 + idx] = out[q];
}

// 64 x 64 tiles while 128 x 128 ones would be fewer than 16 (every decoder weight except the FFN's 1024 x 256)
static int tn_tile(int N, int K) { return ceil_div(N, GT) * ceil_div(K, GT) < 16 ? 64 : GT; }
static int tn_splits(int64_t M, int N, int K, int T, bool bf) {
    const int64_t tiles = ceil_div(N, T) * ceil_div(K, T);
    // workgroups aimed at: ~3 per CU for the fp32 kernels (measured best of 384 / 768 / 1536), 1.5 for the bf16 one (its
    // workgroups are 16x shorter); every extra split costs N*K*4 bytes written and read again by the reduce
    static const int target = [] { const char* e = getenv("U3D_TN_WGS"); return e && atoi(e) > 0 ? atoi(e) : 768; }();
    int64_t s = ceil_div(bf ? 384 : target, tiles);
    const int64_t max_s = ceil_div(M, 4 * GK);
    if (s > max_s) s = max_s;
    return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}

// Tile choice by a small cost model.  All workgroups of these launches are resident at once (<= 5 per CU by LDS), so a launch
// lasts as long as its most loaded CU: ceil(workgroups / 256) workgroup-times -- round 2's 128x64 tiles gave the decoder's
// [16.8k x 256] products 525 workgroups = 2.05 per CU, i.e. three on some CUs and a third of the machine idle behind them
// (0.51 of the MFMA peak).  Candidates 128x128, 128x64, 64x64: time ~ ceil(wgs / 256) * tile area * (1 + overhead of the tile:
// LDS fragment reads and staging per MFMA grow as the tile shrinks).  Measured on the bench step (MI355X, round 3 visit F): all
// GEMMs 6.71 ms with 128x64 everywhere, 6.37 ms with 64x64 everywhere -> the overhead of the small tile is small: 4 % / 8 %.
template <int EPI>
static void launch_nt(const float* A, const float* W, const float* bias, float* C, int64_t M, int N, int K, const float* aux, float* pre,
                      bool bf16_operands, bool x3, hipStream_t s, const void* wplanes = nullptr) {
    static const int force = [] { const char* e = getenv("U3D_NT_TILE"); return e ? atoi(e) : 0; }();       // 1 / 2 / 3 = 128x128 / 128x64 / 64x64
    const int tm[3] = {128, 128, 64}, tn[3] = {128, 64, 64};
    const double over[3] = {1.0, 1.04, 1.08};
    int best = 0;
    double best_t = 0.0;
    for (int c = x3 ? 1 : 0; c < 3; ++c) {         // bf16x3: 128 x 64 at most -- the low-order tile doubles the accumulator registers
        const int64_t wgs = ceil_div(M, tm[c]) * ceil_div(N, tn[c]);
        const double t = (double)ceil_div(wgs, 256) * tm[c] * tn[c] * over[c];
        if (c == (x3 ? 1 : 0) || t < best_t) { best = c; best_t = t; }
    }
    // Round 4: under amdgpu_waves_per_eu(3) the three-plane 128 x 64 kernel takes 162 registers instead of 200 (no spills) and three
    // workgroups share a CU; measured on the bench step (same box, all GEMMs): model's choice 5.39 ms, 128 x 64 everywhere 5.33 ms
    // (with two per CU, round 3: 5.82 ms) -- the model's rounds of 256 workgroups no longer describe it, so large products take it.
    if (x3 && M >= 4096) best = 1;
    if (force >= 1 && force <= 3) best = force - 1;
    if (x3 && best == 0) best = 1;
    const dim3 grid((unsigned)(ceil_div(M, tm[best]) * ceil_div(N, tn[best])));       // (row tile, column tile) decoded in the kernel
#define U3D_NT_LAUNCH(KERNEL)                                                                                                          \
    if (best == 0) hipLaunchKernelGGL((KERNEL<128, EPI, 128>), grid, dim3(256), 0, s, A, W, bias, C, M, N, K, aux, pre);                \
    else if (best == 1) hipLaunchKernelGGL((KERNEL<64, EPI, 128>), grid, dim3(256), 0, s, A, W, bias, C, M, N, K, aux, pre);            \
    else hipLaunchKernelGGL((KERNEL<64, EPI, 64>), grid, dim3(256), 0, s, A, W, bias, C, M, N, K, aux, pre);
    if (bf16_operands) { U3D_NT_LAUNCH(gemm_nt_bf16_k) }
    else if (x3 && wplanes) {
        if (best == 1) hipLaunchKernelGGL((gemm_nt_x3_k<64, EPI, 128, true>), grid, dim3(256), 0, s, A, W, bias, C, M, N, K, aux, pre, wplanes);
        else hipLaunchKernelGGL((gemm_nt_x3_k<64, EPI, 64, true>), grid, dim3(256), 0, s, A, W, bias, C, M, N, K, aux, pre, wplanes);
    } else if (x3) {
        if (best == 1) hipLaunchKernelGGL((gemm_nt_x3_k<64, EPI, 128, false>), grid, dim3(256), 0, s, A, W, bias, C, M, N, K, aux, pre, (const void*)nullptr);
        else hipLaunchKernelGGL((gemm_nt_x3_k<64, EPI, 64, false>), grid, dim3(256), 0, s, A, W, bias, C, M, N, K, aux, pre, (const void*)nullptr);
    } else { U3D_NT_LAUNCH(gemm_nt_k) }
#undef U3D_NT_LAUNCH
}

// epi: 0..4 (see nt_epilogue); + 8: bf16 MFMA operands (fp32 data in HBM, fp32 accumulation)
// The pre-split planes of the NEXT launch's W operand (u3d_gemm_w_planes): per host thread, consumed (and cleared) by the next NT
// entry point called on that thread -- the caller sets them immediately before the call they belong to.
// Each slot remembers the fp32 matrix the planes were made from: an entry point only uses planes whose owner is the W it was handed
// (an exception between the hand-over and the launch must not leave planes behind for an unrelated product).
static thread_local const void* g_wplanes[2] = {nullptr, nullptr};
static thread_local const void* g_wowner[2] = {nullptr, nullptr};
static const void* take_wplanes(int i, const void* W) {
    const void* p = g_wowner[i] == W ? g_wplanes[i] : nullptr;
    g_wplanes[i] = nullptr;
    g_wowner[i] = nullptr;
    return p;
}

static int gemm_nt_epi(const float* A, const float* W, const float* bias, float* C, int64_t M, int N, int K, int epi, const float* aux,
                       float* pre, double flops_hint, hipStream_t s, const void* wplanes = nullptr) {
    const bool bf = (epi & 8) != 0;
    epi &= 7;
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || epi < 0 || epi > 5 || (epi == 2 && !pre) || (epi >= 3 && !aux)) return U3D_EINVAL;
    const bool x3 = !bf && fp32_x3() && K % GKH == 0;        // K = 16 (mod 32) keeps the native fp32 kernel
    if (K % (bf ? GKH : GK)) { set_error("gemm_nt: K=%d must be a multiple of %d", K, bf ? GKH : GK); return U3D_EUNSUPPORTED; }
    ProfScope prof(U3D_K_GEMM, s, flops_hint);
    if ((int64_t)GT * K * 4 >= 0x7fffffffLL || (int64_t)GT * N * 4 >= 0x7fffffffLL) { set_error("gemm_nt: N=%d / K=%d too large for 32-bit tile offsets", N, K); return U3D_EUNSUPPORTED; }
    switch (epi) {
        case 0: launch_nt<0>(A, W, bias, C, M, N, K, aux, pre, bf, x3, s, wplanes); break;
        case 1: launch_nt<1>(A, W, bias, C, M, N, K, aux, pre, bf, x3, s, wplanes); break;
        case 2: launch_nt<2>(A, W, bias, C, M, N, K, aux, pre, bf, x3, s, wplanes); break;
        case 3: launch_nt<3>(A, W, bias, C, M, N, K, aux, pre, bf, x3, s, wplanes); break;
        case 4: launch_nt<4>(A, W, bias, C, M, N, K, aux, pre, bf, x3, s, wplanes); break;
        default: launch_nt<5>(A, W, bias, C, M, N, K, aux, pre, bf, x3, s, wplanes); break;
    }
    return check_launch("gemm_nt");
}

}  // namespace u3d

using namespace u3d;

extern "C" {


int u3d_gemm_nt(const float* A, const float* W, const float* bias, float* C, int64_t M, int N, int K, double flops_hint,
                u3d_stream_t stream) {
    const void* wp = take_wplanes(0, W);
    take_wplanes(1, nullptr);
    return gemm_nt_epi(A, W, bias, C, M, N, K, 0, nullptr, nullptr, flops_hint, (hipStream_t)stream, wp);
}

// `act` of the three entry points below: 0 none, 1 ReLU, 2 GELU; + U3D_BF16_OPERANDS (16) selects the bf16-operand kernels
int u3d_linear_act(const float* X, const float* W, const float* bias, int act, float* pre, float* Y, int64_t M, int N, int K,
                   double flops_hint, u3d_stream_t stream) {
    const int bf = (act & U3D_BF16_OPERANDS) ? 8 : 0;
    act &= ~U3D_BF16_OPERANDS;
    const void* wp = take_wplanes(0, W);
    take_wplanes(1, nullptr);
    if (act < 0 || act > 2) return U3D_EINVAL;
    return gemm_nt_epi(X, W, bias, Y, M, N, K, act | bf, nullptr, pre, flops_hint, (hipStream_t)stream, wp);
}

int u3d_linear_dact(const float* dY, const float* Wt, const float* aux, int act, float* dX, int64_t M, int N, int K, double flops_hint,
                    u3d_stream_t stream) {
    const int bf = (act & U3D_BF16_OPERANDS) ? 8 : 0;
    act &= ~U3D_BF16_OPERANDS;
    const void* wp = take_wplanes(0, Wt);
    take_wplanes(1, nullptr);
    if (act < 0 || act > 2) return U3D_EINVAL;
    return gemm_nt_epi(dY, Wt, nullptr, dX, M, N, K, (act == 0 ? 0 : act + 2) | bf, aux, nullptr, flops_hint, (hipStream_t)stream, wp);
}

int u3d_gemm_nt_add(const float* A, const float* W, const float* addend, int flags, float* C, int64_t M, int N, int K, double flops_hint,
                    u3d_stream_t stream) {
    const void* wp = take_wplanes(0, W);
    take_wplanes(1, nullptr);
    return gemm_nt_epi(A, W, nullptr, C, M, N, K, 5 | ((flags & U3D_BF16_OPERANDS) ? 8 : 0), addend, nullptr, flops_hint, (hipStream_t)stream, wp);
}

int u3d_ln_linear(const float* X, const float* RES, const float* gamma, const float* beta, float eps, float* SUM, float* NQ, float* STATS,
                  const float* W, const float* bias, int act, float* PRE, float* Y, int64_t M, int C, int N, double flops_hint,
                  u3d_stream_t stream) {
    const void* wp = take_wplanes(0, W);
    take_wplanes(1, nullptr);
    if (!NQ || !Y || !W) return U3D_EINVAL;
    int rc = u3d_layer_norm_fwd(X, RES, gamma, beta, M, C, eps, SUM, NQ, STATS, stream);
    if (rc || M == 0) return rc;
    g_wplanes[0] = wp;
    g_wowner[0] = wp ? W : nullptr;
    return u3d_linear_act(NQ, W, bias, act, PRE, Y, M, N, C, flops_hint, stream);
}

int u3d_ffn_fwd(const float* X, const float* W1, const float* b1, const float* W2, const float* b2, int act, float* H, float* A, float* Z,
                int64_t M, int d_in, int hid, int d_out, double flops_hint, u3d_stream_t stream) {
    const void* wp1 = take_wplanes(0, W1);
    const void* wp2 = take_wplanes(1, W2);
    const int bf = (act & U3D_BF16_OPERANDS) ? 8 : 0;
    act &= ~U3D_BF16_OPERANDS;
    if (act != 1 && act != 2) return U3D_EINVAL;
    if (!A || !Z) return U3D_EINVAL;
    const double f1 = flops_hint > 0 ? 2.0 * M * d_in * hid : 0.0, f2 = flops_hint > 0 ? 2.0 * M * hid * d_out : 0.0;
    int rc = gemm_nt_epi(X, W1, b1, A, M, hid, d_in, act | bf, nullptr, H, f1, (hipStream_t)stream, wp1);
    if (rc) return rc;
    return gemm_nt_epi(A, W2, b2, Z, M, d_out, hid, bf, nullptr, nullptr, f2, (hipStream_t)stream, wp2);
}

// three-plane form (gemm_tn_x3_k): 128 x 64 tiles where gemm_tn_k takes 128 x 128 ones, 64 x 64 otherwise
static bool tn_x3_on() { static const int on = [] { const char* e = getenv("U3D_TN_X3"); return e ? atoi(e) : 1; }(); return on != 0 && fp32_x3(); }
static int tn_x3_ta(int N, int K) { return tn_tile(N, K) == GT ? 128 : 64; }
static int tn_x3_splits(int64_t M, int N, int K) {
    const int64_t tiles = ceil_div(N, tn_x3_ta(N, K)) * ceil_div(K, 64);
    int64_t s = ceil_div(768, tiles);
    const int64_t max_s = ceil_div(M, 4 * TXK);
    if (s > max_s) s = max_s;
    return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}

int64_t u3d_gemm_tn_ws_bytes(int64_t M, int N, int K) {
    // the fp32 and the bf16 kernel pick their split counts independently (U3D_TN_WGS moves only the former): size for the larger
    const int s32 = tn_splits(M, N, K, GT, false), s32b = tn_splits(M, N, K, tn_tile(N, K), false), s16 = tn_splits(M, N, K, GT, true);
    const int sx3 = tn_x3_splits(M, N, K);
    int smax = s32 > s16 ? (s32 > s32b ? s32 : s32b) : (s16 > s32b ? s16 : s32b);
    smax = smax > sx3 ? smax : sx3;
    return (int64_t)(smax + 8) * ((int64_t)N * K + N) * 4 + 256;       // + 8: the fp32 grid is padded to whole groups of 8 splits
}

static int gemm_tn_impl(const float* A, const float* B, float* C, float* colsum_A, int64_t M, int N, int K, void* ws, double flops_hint,
                        u3d_stream_t stream, bool bf) {
    if (!A || !B || !C || !ws || M <= 0 || N <= 0 || K <= 0) return U3D_EINVAL;
    if (N % 4 || K % 4) { set_error("gemm_tn: N=%d, K=%d must be multiples of 4", N, K); return U3D_EUNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_GEMM, s, flops_hint);
    // fp32 weight gradients: three-plane products (gemm_tn_x3_k) in the bf16x3 math mode since round 4 -- all GEMM launches of the
    // bench step 5.87 -> 5.46 ms (U3D_TN_X3=0 keeps gemm_tn_k, same box).  Round 3's attempt (pair-packed planes, ds_read_b32
    // fragments, every wave splitting what it read) had 64 x 64 tiles at 45 us against 45 us and 128 x 128 tiles out of registers
    // once the low-order products got a tile of their own; the transpose read and the once-per-element split are what changed.
    const bool x3 = !bf && tn_x3_on();
    const int T = bf ? GT : tn_tile(N, K);
    const int S = x3 ? tn_x3_splits(M, N, K) : tn_splits(M, N, K, T, bf);
    const int64_t rps = x3 ? ceil_div(ceil_div(M, S), TXK) * TXK : ceil_div(ceil_div(M, S), GK) * GK;
    if ((int64_t)(rps + 2 * TXK) * N * 4 >= 0x7fffffffLL || (int64_t)(rps + 2 * TXK) * K * 4 >= 0x7fffffffLL || (int64_t)N * K * 4 >= 0x7fffffffLL) {
        set_error("gemm_tn: M=%lld N=%d K=%d too large for 32-bit split offsets", (long long)M, N, K);
        return U3D_EUNSUPPORTED;
    }
    if (x3) {
        const int ta = tn_x3_ta(N, K);
        const unsigned grid = (unsigned)(ceil_div(S, 8) * 8 * ceil_div(N, ta) * ceil_div(K, 64));
        if (ta == 128) hipLaunchKernelGGL((gemm_tn_x3_k<128, 64>), dim3(grid), dim3(256), 0, s, A, B, (float*)ws, colsum_A ? 1 : 0, M, N, K, rps, S);
        else hipLaunchKernelGGL((gemm_tn_x3_k<64, 64>), dim3(grid), dim3(256), 0, s, A, B, (float*)ws, colsum_A ? 1 : 0, M, N, K, rps, S);
    } else if (bf) hipLaunchKernelGGL(gemm_tn_bf16_k, dim3((unsigned)ceil_div(N, GT), (unsigned)ceil_div(K, GT), S), dim3(256), 0, s, A, B, (float*)ws, colsum_A ? 1 : 0, M, N, K, rps);
    else {
        const unsigned grid = (unsigned)(ceil_div(S, 8) * 8 * ceil_div(N, T) * ceil_div(K, T));       // whole groups of 8 splits
        if (T == 64) hipLaunchKernelGGL(gemm_tn_k<64>, dim3(grid), dim3(256), 0, s, A, B, (float*)ws, colsum_A ? 1 : 0, M, N, K, rps, S);
        else hipLaunchKernelGGL(gemm_tn_k<GT>, dim3(grid), dim3(256), 0, s, A, B, (float*)ws, colsum_A ? 1 : 0, M, N, K, rps, S);
    }
    const int64_t n4_main = (int64_t)N * K / 4, n4 = n4_main + (colsum_A ? N / 4 : 0);
    int64_t grid = ceil_div(n4, 256);
    grid = grid > 1024 ? 1024 : grid;
    hipLaunchKernelGGL(gemm_tn_reduce_k, dim3((unsigned)grid), dim3(256), 0, s, (const float*)ws, S, n4, n4_main, C, colsum_A);
    return check_launch("gemm_tn");
}

int u3d_gelu_fwd(const float* h, float* a, int64_t n, u3d_stream_t stream) {
    if (!h || !a || n <= 0 || n % 4) return U3D_EINVAL;
    int64_t g = ceil_div(n / 4, 256);
    hipLaunchKernelGGL(gelu_fwd_k, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, (hipStream_t)stream, (const float4*)h, (float4*)a, n / 4);
    return check_launch("gelu_fwd");
}

int u3d_gelu_bwd(const float* da, const float* h, float* dh, int64_t n, u3d_stream_t stream) {
    if (!da || !h || !dh || n <= 0 || n % 4) return U3D_EINVAL;
    int64_t g = ceil_div(n / 4, 256);
    hipLaunchKernelGGL(gelu_bwd_k, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, (hipStream_t)stream, (const float4*)da, (const float4*)h, (float4*)dh, n / 4);
    return check_launch("gelu_bwd");
}

int u3d_gemm_tn(const float* A, const float* B, float* C, float* colsum_A, int64_t M, int N, int K, void* ws, double flops_hint,
                u3d_stream_t stream) {
    return gemm_tn_impl(A, B, C, colsum_A, M, N, K, ws, flops_hint, stream, false);
}

int u3d_gemm_tn_bf16(const float* A, const float* B, float* C, float* colsum_A, int64_t M, int N, int K, void* ws, double flops_hint,
                     u3d_stream_t stream) {
    return gemm_tn_impl(A, B, C, colsum_A, M, N, K, ws, flops_hint, stream, true);
}

int u3d_transpose(const float* in, float* out, int R, int C, u3d_stream_t stream) {
    if (!in || !out || R <= 0 || C <= 0) return U3D_EINVAL;
    hipLaunchKernelGGL(transpose_k, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, (hipStream_t)stream, in, out, R, C);
    return check_launch("transpose");
}

int u3d_weight_planes_batch(const void* desc, int n_desc, int64_t total_blocks, u3d_stream_t stream) {
    if (!desc || n_desc <= 0 || total_blocks <= 0 || total_blocks >= 0x7fffffffLL) return U3D_EINVAL;
    hipLaunchKernelGGL(planes_batch_k, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const PlDesc*)desc, n_desc);
    return check_launch("weight_planes_batch");
}

int u3d_gemm_w_planes(const void* w, const void* planes, const void* w2, const void* planes2) {
    g_wplanes[0] = planes; g_wowner[0] = planes ? w : nullptr;
    g_wplanes[1] = planes2; g_wowner[1] = planes2 ? w2 : nullptr;
    return U3D_OK;
}

int u3d_transpose_batch(const void* desc, int n_desc, int64_t total_blocks, u3d_stream_t stream) {
    if (!desc || n_desc <= 0 || total_blocks <= 0 || total_blocks >= 0x7fffffffLL) return U3D_EINVAL;
    hipLaunchKernelGGL(transpose_batch_k, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const TrDesc*)desc, n_desc);
    return check_launch("transpose_batch");
}

}  // extern "C"
