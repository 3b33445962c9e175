// K14b: the decoder's Linear layers with bf16 ACTIVATIONS in HBM (BASELINE configs[2]; the reference's `--amp` run autocasts the
// Linear / MultiheadAttention activations to 16 bits, tools/train.py:86-99 around unidet3d/encoder.py:19-21,55-61,138-163).
// gemm.hip's bf16-operand kernels read fp32 tensors and round them on the way into LDS: every activation crosses HBM at 4 bytes an
// element although only 8 mantissa bits enter the product (cfg3, round 5: 14.3 GB of GEMM traffic per step).  Here the streamed
// operands and the results may BE bf16 tensors:
//   gemm_nt_b16_k   C[M,N] = A[M,K] . W[N,K]^T (+ bias, ReLU, ReLU mask, addend): A bf16 or fp32, C (and the mask) bf16 or fp32,
//                   W stays fp32 (a weight is <= 1 MB, re-read from L2, rounded while it is staged);
//   gemm_tn_b16_k   partial[N,K] = A[M,N]^T . B[M,K] (weight gradients): either operand bf16 or fp32, fp32 partials + the fixed-order
//                   reduce of gemm.hip;
//   gelu_*_b16_k    erf GELU between two bf16 tensors (the FFN's hidden activation never exists in fp32).
// A bf16 operand is staged with 16-byte loads (8 values) and 16-byte LDS writes and no conversion arithmetic at all; a bf16 result
// leaves through 4-byte stores: lanes 2j and 2j+1 of the 32 x 32 accumulator hold neighbouring columns, so they trade one value per
// row pair (DPP quad_perm) and each stores two columns of one row -- half the store instructions of the fp32 epilogue, half the bytes.
// Values: a bf16 tensor written by one kernel and read by the next holds exactly the value gemm.hip's kernels would have formed by
// rounding the fp32 tensor in flight (round to nearest even both ways), so the products are unchanged; what changes is that
// ELEMENTWISE consumers (GELU and its derivative) see the rounded pre-activation and hand on a rounded gradient.
#include "u3d_common.h"

namespace u3d {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

constexpr int HT = 128;        // macro tile
constexpr int HK = 32;         // K-step
constexpr int HL = HK + 8;     // padded LDS row of the NT tiles (halves): 80-byte rows, conflict-free 16-byte reads of 32 rows

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) { return __builtin_bit_cast(unsigned, bf16x2{(__bf16)lo, (__bf16)hi}); }

// erf GELU with one exponential (gemm.hip gelu_parts: Abramowitz-Stegun 7.1.26)
__device__ __forceinline__ void gelu_parts16(float x, float& cdf, float& pdf) {
    const float ax = fabsf(x), e = __expf(-0.5f * x * x);
    const float t = __frcp_rn(1.f + 0.3275911f * 0.70710678118654752440f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float tail = 0.5f * poly * e;
    cdf = x >= 0.f ? 1.f - tail : tail;
    pdf = 0.39894228040143267794f * e;
}

#ifndef U3D_NT16_ABL
#define U3D_NT16_ABL 0          // timing ablations (wrong results): 1 no result stores, 2 no MFMAs, 4 no global loads in the loop, 8 no LDS stores in the loop
#endif
// EPI: 0 C = acc + bias | 1 C = relu(acc + bias) | 3 C = aux > 0 ? acc : 0 (aux = the ReLU output, C's dtype) | 5 C = acc + aux (fp32 both)
template <int TN, int EPI, int TM, bool A16, bool C16>
__global__ __launch_bounds__(256) void gemm_nt_b16_k(const void* __restrict__ A_, const float* __restrict__ W, const float* __restrict__ bias,
                                                     void* __restrict__ C_, int64_t M, int N, int K, const void* __restrict__ aux_) {
    constexpr int NB = TN / 64, TA = TM / 64;
    constexpr int EA = A16 ? 2 : 4, EC = C16 ? 2 : 4;
    __shared__ __attribute__((aligned(16))) __bf16 As[2][TM * HL];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][TN * HL];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, i32 = lane & 31, kh = lane >> 5;
    const int nt_ = (N + TN - 1) / TN;                       // XCD-aware 1-D grid (gemm.hip gemm_nt_k): the column tiles of a row tile share an L2
    const int64_t wid_ = xcd_swizzle(blockIdx.x, gridDim.x);
    const int64_t m0 = (wid_ / nt_) * TM;
    const int n0 = (int)(wid_ % nt_) * TN;
    const int rows_a = (int)min((int64_t)TM, M - m0), rows_b = min(TN, N - n0);
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(reinterpret_cast<const char*>(A_) + m0 * K * EA, (int64_t)rows_a * K * EA);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(W + (int64_t)n0 * K, (int64_t)rows_b * K * 4);
    // staging map: thread -> (row = tid >> 2 (+ 64 j), the 8 values at column 8 * (tid & 3)): 32 bytes of fp32 or 16 bytes of bf16
    const int srow = tid >> 2, sc8 = tid & 3;
    const int voa = (srow * K + sc8 * 8) * EA, vsa = 64 * K * EA;
    const int vob = (srow * K + sc8 * 8) * 4, vsb = 64 * K * 4;
    f32x4 ra[TA][A16 ? 1 : 2], rb[NB][2];
    const int nk = K / HK;
    auto gload = [&](int kt) {
#pragma unroll
        for (int j = 0; j < TA; ++j) {
            ra[j][0] = bload128(rs_a, voa + j * vsa, kt * (HK * EA));
            if constexpr (!A16) ra[j][1] = bload128(rs_a, voa + j * vsa + 16, kt * (HK * EA));
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            rb[j][0] = bload128(rs_b, vob + j * vsb, kt * (HK * 4));
            rb[j][1] = bload128(rs_b, vob + j * vsb + 16, kt * (HK * 4));
        }
    };
    auto cvt8 = [](const f32x4& lo, const f32x4& hi) {
        return bf16x8{(__bf16)lo[0], (__bf16)lo[1], (__bf16)lo[2], (__bf16)lo[3], (__bf16)hi[0], (__bf16)hi[1], (__bf16)hi[2], (__bf16)hi[3]};
    };
    // Loads run TWO k-steps ahead of their use (a workgroup's life is mostly load latency: tools/prof_gemm16.py ablations -- without the
    // loop's loads 25 instead of 37 us at N, K = 256, 1024, without MFMAs 36): the rows of step kt+1, loaded during step kt-1, move from
    // the load registers to the staging registers `sa` / `sb` (the fp32 ones are rounded here) after the products of step kt, the loads
    // of step kt+2 are issued into the freed registers, and the staging registers go to the other LDS buffer before the step's barrier.
    bf16x8 sa[TA], sb[NB];
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < TA; ++j) {
            if constexpr (A16) sa[j] = __builtin_bit_cast(bf16x8, ra[j][0]);
            else sa[j] = cvt8(ra[j][0], ra[j][1]);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) sb[j] = cvt8(rb[j][0], rb[j][1]);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < TA; ++j) *reinterpret_cast<bf16x8*>(&As[buf][(srow + 64 * j) * HL + sc8 * 8]) = sa[j];
#pragma unroll
        for (int j = 0; j < NB; ++j) *reinterpret_cast<bf16x8*>(&Bs[buf][(srow + 64 * j) * HL + sc8 * 8]) = sb[j];
    };
    f32x16 acc[TA][NB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    auto compute = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 af[TA], bf[NB];
#pragma unroll
            for (int t = 0; t < TA; ++t) af[t] = *reinterpret_cast<const bf16x8*>(&As[buf][(wr * (TM / 2) + t * 32 + i32) * HL + h * 16 + kh * 8]);
#pragma unroll
            for (int t = 0; t < NB; ++t) bf[t] = *reinterpret_cast<const bf16x8*>(&Bs[buf][(wc * (TN / 2) + t * 32 + i32) * HL + h * 16 + kh * 8]);
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if constexpr (U3D_NT16_ABL & 2) acc[a][b][0] += (float)af[a][0] + (float)bf[b][1];
                    else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
                }
        }
    };
    // The main loop has NO branch around its loads (a conditional load makes the compiler copy the loop-carried load registers behind
    // a vmcnt(0), which serialises the loop) and never loads out of range: the last two steps, whose two-ahead loads do not exist,
    // are peeled off (measured ahead of loading "steps past the end" through an out-of-range offset by 4-8 %).  The LDS
    // stores come BEFORE the next loads: a bf16 operand's staging registers are its load registers, and while they are live across
    // the load the compiler gives the load new registers and copies them back behind a vmcnt wait at the loop's end.
    gload(0);
    stage();
    lstore(0);
    if (nk > 1) gload(1);
    __syncthreads();
    int kt = 0;
    for (; kt + 2 < nk; ++kt) {
        const int buf = kt & 1;
        compute(buf);
        stage();                                                       // step kt+1 (loaded one step ago)
        if constexpr (!(U3D_NT16_ABL & 8)) lstore(buf ^ 1);             // every wave left that buffer at the barrier of step kt-1
        if constexpr (!(U3D_NT16_ABL & 4)) gload(kt + 2);
        __syncthreads();
    }
    if (kt + 1 < nk) {                                                 // step nk-2: stages the last step, nothing left to load
        compute(kt & 1);
        stage();
        lstore((kt & 1) ^ 1);
        __syncthreads();
        ++kt;
    }
    compute(kt & 1);                                                   // step nk-1
    // ---- epilogue.  D layout of the 32 x 32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    if constexpr (U3D_NT16_ABL & 1) { if (acc[0][0][0] != 12345.678f) return; }
    char* Cb = reinterpret_cast<char*>(C_) + m0 * N * EC;
    const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(Cb, (int64_t)rows_a * N * EC);
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(reinterpret_cast<const char*>(aux_) + (EPI >= 3 ? m0 * N * EC : 0), (int64_t)rows_a * N * EC);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int n = n0 + wc * (TN / 2) + b * 32 + i32;
        const float bv = (bias && n < N) ? bias[n] : 0.f;
        if constexpr (!C16) {
            const int vc = n < N ? (4 * kh * N + n) * 4 : 0x7fffffff;          // columns past N: dropped by the bounds check
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wr * (TM / 2) + a * 32 + (r & 3) + 8 * (r >> 2);
                    float v = acc[a][b][r] + bv;
                    if constexpr (EPI == 1) v = fmaxf(v, 0.f);
                    if constexpr (EPI == 3 || EPI == 5) {
                        const float x = __builtin_bit_cast(float, bload32(rs_x, vc, row * N * 4));
                        v = EPI == 3 ? (x > 0.f ? v : 0.f) : v + x;
                    }
                    asm volatile("" : "+v"(v));
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_c, vc, row * N * 4, 0);
                }
        } else {
            // rows come in neighbouring pairs (r, r + 1), r even; the even lane of a lane pair takes row r, the odd lane row r + 1, both
            // columns (n & ~1, n | 1): one value crosses to the partner lane, each lane rounds two values and stores 4 bytes (N is even)
            const int odd = lane & 1, ne = n & ~1;
            const int vc = ne < N ? ((4 * kh + odd) * N + ne) * 2 : 0x7fffffff;
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int rp = 0; rp < 8; ++rp) {
                    const int r0 = 2 * rp, row = wr * (TM / 2) + a * 32 + (r0 & 3) + 8 * (r0 >> 2);       // row of the EVEN lane; the odd lane's is row + 1 (in vc)
                    float v0 = acc[a][b][r0] + bv, v1 = acc[a][b][r0 + 1] + bv;
                    if constexpr (EPI == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                    const float send = odd ? v0 : v1;
                    const float recv = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
                    float lo = odd ? recv : v0, hi = odd ? v1 : recv;           // columns ne, ne + 1 of this lane's row
                    if constexpr (EPI == 3) {
                        const unsigned x = (unsigned)bload32(rs_x, vc, row * N * 2);
                        lo = bf16_bits_to_f32((unsigned short)(x & 0xffffu)) > 0.f ? lo : 0.f;
                        hi = bf16_bits_to_f32((unsigned short)(x >> 16)) > 0.f ? hi : 0.f;
                    }
                    unsigned w = pack_bf16(lo, hi);
                    asm volatile("" : "+v"(w));
                    __builtin_amdgcn_raw_buffer_store_b32(w, rs_c, vc, row * N * 2, 0);
                }
        }
    }
}

// ---- weight gradient: partial[s][n][k] = sum over the split's rows of A[m][n] B[m][k]; 128 x 128 output tile, 32 rows per stage.
// The reduction index is the ROW of both operands: a lane's 8 k values are a column of the staged [32 rows][128 cols] tile, two
// ds_read_b64_tr_b16 per fragment (gemm.hip gemm_tn_bf16_k).  A bf16 operand is staged by 16-byte copies: thread -> (row = tid >> 4
// (+ 16), 8 columns at 8 (tid & 15)); an fp32 one as in gemm.hip: (row = tid >> 5 (+ 8 j), 4 columns at 4 (tid & 31)), rounded here.
constexpr int WKH = 32;
constexpr int WLH = HT + 32;         // padded LDS row (halves): the four rows a half-wave reads lie on four 64-byte bank groups

template <bool A16, bool B16>
__global__ __launch_bounds__(256) void gemm_tn_b16_k(const void* __restrict__ A_, const void* __restrict__ B_, float* __restrict__ partial,
                                                     int colsum, int64_t M, int N, int K, int64_t rows_per_split, int S) {
    __shared__ __attribute__((aligned(16))) __bf16 As[2][WKH * WLH];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][WKH * WLH];
    __shared__ float csum_s[16][HT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, i32 = lane & 31, kh = lane >> 5;
    // 1-D grid, XCD-aware (gemm.hip gemm_tn_k): workgroup b runs on XCD b % 8, so the tiles of ONE row split are made neighbours on
    // one XCD -- its rows of A and B come from HBM once and from that XCD's L2 for the other tiles.  (The 3-D grid of the first version
    // spread a split's 16 tiles over all eight L2s: PMC 175 MB fetched per launch for 63 MB of bf16 operands.)
    const int tiles_n = (N + HT - 1) / HT, tiles = tiles_n * ((K + HT - 1) / HT);
    const int slot = blockIdx.x >> 3, tile = slot % tiles;
    const int split = (slot / tiles) * 8 + (blockIdx.x & 7);
    if (split >= S) return;
    const int n0 = (tile % tiles_n) * HT, k0 = (tile / tiles_n) * HT;
    const int64_t mlo = (int64_t)split * rows_per_split;
    const int64_t mhi = min(M, mlo + rows_per_split);
    const int rows = (int)max((int64_t)0, mhi - mlo);
    constexpr int EA = A16 ? 2 : 4, EB = B16 ? 2 : 4;
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(reinterpret_cast<const char*>(A_) + mlo * N * EA, (int64_t)rows * N * EA);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(reinterpret_cast<const char*>(B_) + mlo * K * EB, (int64_t)rows * K * EB);
    // fp32 map
    const int srow = tid >> 5, sc4 = tid & 31;
    // bf16 map
    const int hrow = tid >> 4, hc8 = tid & 15;
    const int va = A16 ? (n0 + hc8 * 8 < N ? (hrow * N + n0 + hc8 * 8) * 2 : 0x7fffffff) : (n0 + sc4 * 4 < N ? (srow * N + n0 + sc4 * 4) * 4 : 0x7fffffff);
    const int vb = B16 ? (k0 + hc8 * 8 < K ? (hrow * K + k0 + hc8 * 8) * 2 : 0x7fffffff) : (k0 + sc4 * 4 < K ? (srow * K + k0 + sc4 * 4) * 4 : 0x7fffffff);
    const bool ca = va != 0x7fffffff, cb = vb != 0x7fffffff;
    constexpr int NLA = A16 ? 2 : 4, NLB = B16 ? 2 : 4;
    f32x4 ra[NLA], rb[NLB];
    float cs[A16 ? 8 : 4];
#pragma unroll
    for (int c = 0; c < (A16 ? 8 : 4); ++c) cs[c] = 0.f;
    const int nt = (rows + WKH - 1) / WKH;
    auto gload = [&](int t) {
#pragma unroll
        for (int j = 0; j < NLA; ++j) ra[j] = bload128(rs_a, ca ? va + j * (A16 ? 16 : 8) * N * EA : va, t * (WKH * N * EA));
#pragma unroll
        for (int j = 0; j < NLB; ++j) rb[j] = bload128(rs_b, cb ? vb + j * (B16 ? 16 : 8) * K * EB : vb, t * (WKH * K * EB));
    };
    // staging registers (rounded where the operand is fp32): the loads run two trips ahead of their use, as in gemm_nt_b16_k
    f32x4 sa16[A16 ? NLA : 1], sb16[B16 ? NLB : 1];
    bf16x4 sa32[A16 ? 1 : NLA], sb32[B16 ? 1 : NLB];
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < NLA; ++j) {
            if constexpr (A16) {
                const u32x4 w = __builtin_bit_cast(u32x4, ra[j]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    cs[2 * c] += __builtin_bit_cast(float, w[c] << 16);
                    cs[2 * c + 1] += __builtin_bit_cast(float, w[c] & 0xffff0000u);
                }
                sa16[j] = ra[j];
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) cs[c] += ra[j][c];
                sa32[j] = bf16x4{(__bf16)ra[j][0], (__bf16)ra[j][1], (__bf16)ra[j][2], (__bf16)ra[j][3]};
            }
        }
#pragma unroll
        for (int j = 0; j < NLB; ++j) {
            if constexpr (B16) sb16[j] = rb[j];
            else sb32[j] = bf16x4{(__bf16)rb[j][0], (__bf16)rb[j][1], (__bf16)rb[j][2], (__bf16)rb[j][3]};
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NLA; ++j) {
            if constexpr (A16) *reinterpret_cast<f32x4*>(&As[buf][(hrow + 16 * j) * WLH + hc8 * 8]) = sa16[j];
            else *reinterpret_cast<bf16x4*>(&As[buf][(srow + 8 * j) * WLH + sc4 * 4]) = sa32[j];
        }
#pragma unroll
        for (int j = 0; j < NLB; ++j) {
            if constexpr (B16) *reinterpret_cast<f32x4*>(&Bs[buf][(hrow + 16 * j) * WLH + hc8 * 8]) = sb16[j];
            else *reinterpret_cast<bf16x4*>(&Bs[buf][(srow + 8 * j) * WLH + sc4 * 4]) = sb32[j];
        }
    };
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int troff = ((lane & 15) >> 2) * WLH + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);      // this lane's 8 bytes of its group's [4 rows][16 cols] block
    auto colfrag = [&](const __bf16* t, int row0, int col32) {       // column col32 + i32 over rows row0 .. row0 + 7
        const __bf16* p = t + row0 * WLH + col32 + troff;
        const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
        const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * WLH));
        return __builtin_bit_cast(bf16x8, s16x8{r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]});
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    auto compute = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r0 = h * 16 + kh * 8;
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
    };
    // main loop without a branch around its loads and without out-of-range loads; the last two trips are peeled off (gemm_nt_b16_k).
    // (Rows past the end of the split inside a trip are past the end of the descriptors -- a VECTOR offset: zeros.)
    gload(0);
    stage();
    lstore(0);
    if (nt > 1) gload(1);
    __syncthreads();
    int t = 0;
    for (; t + 2 < nt; ++t) {
        const int buf = t & 1;
        compute(buf);
        stage();                                           // trip t+1 (loaded one trip ago)
        lstore(buf ^ 1);
        gload(t + 2);
        __syncthreads();
    }
    if (t + 1 < nt) {
        compute(t & 1);
        stage();
        lstore((t & 1) ^ 1);
        __syncthreads();
        ++t;
    }
    compute(t & 1);
    const int64_t pstride = (int64_t)N * K + (colsum ? N : 0);          // a split's block: [N*K] products, then [N] column sums
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(partial + (int64_t)split * pstride, (int64_t)N * K * 4);
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
    if (colsum && k0 == 0) {               // column sums of A (the bias gradient): row slots per column -> one value per column, fixed order
        constexpr int SL = A16 ? 16 : 8;
        if constexpr (A16) {
#pragma unroll
            for (int c = 0; c < 8; ++c) csum_s[hrow][hc8 * 8 + c] = cs[c];
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) csum_s[srow][sc4 * 4 + c] = cs[c];
        }
        __syncthreads();
        if (tid < HT && n0 + tid < N) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < SL; ++r) v += csum_s[r][tid];
            partial[(int64_t)split * pstride + (int64_t)N * K + n0 + tid] = v;
        }
    }
}

// sums the splits in a fixed order (gemm.hip gemm_tn_reduce_k: same order, same four independent partial sums)
__global__ __launch_bounds__(256) void gemm_tn_reduce16_k(const float* __restrict__ partial, int S, int64_t n4, int64_t n4_main, float* __restrict__ C,
                                                          float* __restrict__ C2) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
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

// ---- erf GELU between bf16 tensors (8 values per thread): a = gelu(h); dh = da * gelu'(h); fp32 arithmetic, one rounding on the way out
__global__ __launch_bounds__(256) void gelu_fwd_b16_k(const u32x4* __restrict__ h, u32x4* __restrict__ a, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const u32x4 v = h[i];
        u32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float x0 = __builtin_bit_cast(float, v[c] << 16), x1 = __builtin_bit_cast(float, v[c] & 0xffff0000u);
            float c0, p0, c1, p1;
            gelu_parts16(x0, c0, p0);
            gelu_parts16(x1, c1, p1);
            o[c] = pack_bf16(x0 * c0, x1 * c1);
        }
        a[i] = o;
    }
}
__global__ __launch_bounds__(256) void gelu_bwd_b16_k(const u32x4* __restrict__ da, const u32x4* __restrict__ h, u32x4* __restrict__ dh, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const u32x4 g = da[i], v = h[i];
        u32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float x0 = __builtin_bit_cast(float, v[c] << 16), x1 = __builtin_bit_cast(float, v[c] & 0xffff0000u);
            const float g0 = __builtin_bit_cast(float, g[c] << 16), g1 = __builtin_bit_cast(float, g[c] & 0xffff0000u);
            float c0, p0, c1, p1;
            gelu_parts16(x0, c0, p0);
            gelu_parts16(x1, c1, p1);
            o[c] = pack_bf16(g0 * (c0 + x0 * p0), g1 * (c1 + x1 * p1));
        }
        dh[i] = o;
    }
}

template <int EPI, bool A16, bool C16>
static void launch_nt16(const void* A, const float* W, const float* bias, void* C, int64_t M, int N, int K, const void* aux, hipStream_t s) {
    // tile choice: the largest tile that still gives every CU a workgroup.  gemm.hip's rounds-of-256-workgroups model does not
    // describe this kernel (three to four workgroups share a CU and most of a workgroup's life is load latency): measured at
    // M = 24 600 (tools/prof_gemm16.py, us, 128 x 128 | 128 x 64 | 64 x 64): N, K = 256, 256: 10.9 | 11.6 | 12.8; 768, 256: 24.6 | 28.6 |
    // 35.0; 1024, 256: 29.2 | 35.9 | 40.3; 256, 1024: 25.3 | 29.5 | 35.9
    static const int force = [] { const char* e = getenv("U3D_NT16_TILE"); return e ? atoi(e) : 0; }();      // 1 / 2 / 3 = 128x128 / 128x64 / 64x64
    const int tm[3] = {128, 128, 64}, tn[3] = {128, 64, 64};
    int best = 2;
    for (int c = 0; c < 3; ++c)
        if (ceil_div(M, tm[c]) * ceil_div(N, tn[c]) >= 256) { best = c; break; }
    if (N <= 64 && best == 0) best = 1;               // narrow results (class / box heads): no empty half tile
    if (force >= 1 && force <= 3) best = force - 1;
    const dim3 grid((unsigned)(ceil_div(M, tm[best]) * ceil_div(N, tn[best])));
    if (best == 0) hipLaunchKernelGGL((gemm_nt_b16_k<128, EPI, 128, A16, C16>), grid, dim3(256), 0, s, A, W, bias, C, M, N, K, aux);
    else if (best == 1) hipLaunchKernelGGL((gemm_nt_b16_k<64, EPI, 128, A16, C16>), grid, dim3(256), 0, s, A, W, bias, C, M, N, K, aux);
    else hipLaunchKernelGGL((gemm_nt_b16_k<64, EPI, 64, A16, C16>), grid, dim3(256), 0, s, A, W, bias, C, M, N, K, aux);
}

template <int EPI>
static void launch_nt16_types(const void* A, const float* W, const float* bias, void* C, int64_t M, int N, int K, const void* aux, int flags, hipStream_t s) {
    const bool a16 = flags & U3D_A_BF16, c16 = flags & U3D_C_BF16;
    if constexpr (EPI == 5) {          // the addend form is fp32 out only
        if (a16) launch_nt16<EPI, true, false>(A, W, bias, C, M, N, K, aux, s);
        else launch_nt16<EPI, false, false>(A, W, bias, C, M, N, K, aux, s);
    } else {
        if (a16 && c16) launch_nt16<EPI, true, true>(A, W, bias, C, M, N, K, aux, s);
        else if (a16) launch_nt16<EPI, true, false>(A, W, bias, C, M, N, K, aux, s);
        else if (c16) launch_nt16<EPI, false, true>(A, W, bias, C, M, N, K, aux, s);
        else launch_nt16<EPI, false, false>(A, W, bias, C, M, N, K, aux, s);
    }
}

// row splits: about one workgroup per CU, two for the 16-tile products (tools/prof_gemm16.py at M = 24 600, us for 256 | 512 workgroups,
// bf16 operands, XCD-aware grid: N, K = 256, 256: 19.5 | 24.7; 768, 256: 29.8 | 29.5; 1024, 256: 31.8 | 28.9; 256, 1024: 30.4 | 28.1;
// 256, 32: 19.5 | 26.4) -- every split writes an [N, K] fp32 block that the reduce reads back
static int tn16_splits(int64_t M, int N, int K) {
    const int64_t tiles = ceil_div(N, HT) * ceil_div(K, HT);
    static const int forced = [] { const char* e = getenv("U3D_TN16_WGS"); return e && atoi(e) > 0 ? atoi(e) : 0; }();
    const int target = forced ? forced : (tiles >= 16 ? 512 : 256);
    int64_t s = ceil_div(target, tiles);
    const int64_t max_s = ceil_div(M, 4 * WKH);
    if (s > max_s) s = max_s;
    return (int)(s < 1 ? 1 : (s > 256 ? 256 : s));
}

}  // namespace u3d

using namespace u3d;

extern "C" {

int u3d_gemm_nt_b16(const void* A, const float* W, const float* bias, int epi, const void* aux, void* C, int flags, int64_t M, int N, int K,
                    double flops_hint, u3d_stream_t stream) {
    if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0 || (epi != 0 && epi != 1 && epi != 3 && epi != 5) || (epi >= 3 && !aux)) return U3D_EINVAL;
    if (flags & ~(U3D_A_BF16 | U3D_C_BF16)) return U3D_EINVAL;
    if (K % HK) { set_error("gemm_nt_b16: K=%d must be a multiple of %d", K, HK); return U3D_EUNSUPPORTED; }
    if ((flags & U3D_C_BF16) && (N % 2 || epi == 5)) { set_error("gemm_nt_b16: a bf16 result needs an even N (%d) and epi != 5", N); return U3D_EUNSUPPORTED; }
    if ((int64_t)HT * K * 4 >= 0x7fffffffLL || (int64_t)HT * N * 4 >= 0x7fffffffLL) { set_error("gemm_nt_b16: N=%d / K=%d too large for 32-bit tile offsets", N, K); return U3D_EUNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_GEMM, s, flops_hint);
    switch (epi) {
        case 0: launch_nt16_types<0>(A, W, bias, C, M, N, K, aux, flags, s); break;
        case 1: launch_nt16_types<1>(A, W, bias, C, M, N, K, aux, flags, s); break;
        case 3: launch_nt16_types<3>(A, W, bias, C, M, N, K, aux, flags, s); break;
        default: launch_nt16_types<5>(A, W, bias, C, M, N, K, aux, flags, s); break;
    }
    return check_launch("gemm_nt_b16");
}

int64_t u3d_gemm_tn_b16_ws_bytes(int64_t M, int N, int K) { return (int64_t)(tn16_splits(M, N, K) + 1) * ((int64_t)N * K + N) * 4 + 256; }

int u3d_gemm_tn_b16(const void* A, const void* B, float* C, float* colsum_A, int flags, int64_t M, int N, int K, void* ws, double flops_hint,
                    u3d_stream_t stream) {
    if (!A || !B || !C || !ws || M <= 0 || N <= 0 || K <= 0 || (flags & ~(U3D_A_BF16 | U3D_B_BF16))) return U3D_EINVAL;
    const bool a16 = flags & U3D_A_BF16, b16 = flags & U3D_B_BF16;
    if (N % (a16 ? 8 : 4) || K % (b16 ? 8 : 4)) { set_error("gemm_tn_b16: N=%d, K=%d must be multiples of 4 (fp32 operand) / 8 (bf16 operand)", N, K); return U3D_EUNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_GEMM, s, flops_hint);
    const int S = tn16_splits(M, N, K);
    const int64_t rps = ceil_div(ceil_div(M, S), WKH) * WKH;
    if ((int64_t)(rps + 2 * WKH) * N * 4 >= 0x7fffffffLL || (int64_t)(rps + 2 * WKH) * K * 4 >= 0x7fffffffLL || (int64_t)N * K * 4 >= 0x7fffffffLL) {
        set_error("gemm_tn_b16: M=%lld N=%d K=%d too large for 32-bit split offsets", (long long)M, N, K);
        return U3D_EUNSUPPORTED;
    }
    const dim3 grid((unsigned)(ceil_div(S, 8) * 8 * ceil_div(N, HT) * ceil_div(K, HT)));          // whole groups of 8 splits
    const int cs = colsum_A ? 1 : 0;
    if (a16 && b16) hipLaunchKernelGGL((gemm_tn_b16_k<true, true>), grid, dim3(256), 0, s, A, B, (float*)ws, cs, M, N, K, rps, S);
    else if (a16) hipLaunchKernelGGL((gemm_tn_b16_k<true, false>), grid, dim3(256), 0, s, A, B, (float*)ws, cs, M, N, K, rps, S);
    else if (b16) hipLaunchKernelGGL((gemm_tn_b16_k<false, true>), grid, dim3(256), 0, s, A, B, (float*)ws, cs, M, N, K, rps, S);
    else hipLaunchKernelGGL((gemm_tn_b16_k<false, false>), grid, dim3(256), 0, s, A, B, (float*)ws, cs, M, N, K, rps, S);
    const int64_t n4_main = (int64_t)N * K / 4, n4 = n4_main + (colsum_A ? N / 4 : 0);
    int64_t g = ceil_div(n4, 256);
    g = g > 1024 ? 1024 : g;
    hipLaunchKernelGGL(gemm_tn_reduce16_k, dim3((unsigned)g), dim3(256), 0, s, (const float*)ws, S, n4, n4_main, C, colsum_A);
    return check_launch("gemm_tn_b16");
}

int u3d_gelu_fwd_b16(const void* h, void* a, int64_t n, u3d_stream_t stream) {
    if (!h || !a || n <= 0 || n % 8) return U3D_EINVAL;
    int64_t g = ceil_div(n / 8, 256);
    hipLaunchKernelGGL(gelu_fwd_b16_k, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)h, (u32x4*)a, n / 8);
    return check_launch("gelu_fwd_b16");
}

int u3d_gelu_bwd_b16(const void* da, const void* h, void* dh, int64_t n, u3d_stream_t stream) {
    if (!da || !h || !dh || n <= 0 || n % 8) return U3D_EINVAL;
    int64_t g = ceil_div(n / 8, 256);
    hipLaunchKernelGGL(gelu_bwd_b16_k, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)da, (const u32x4*)h, (u32x4*)dh, n / 8);
    return check_launch("gelu_bwd_b16");
}

}  // extern "C"
