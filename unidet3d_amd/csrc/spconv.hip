// K4-K8 sparse convolutions on gfx950: one output-stationary gather-MFMA-scatter kernel for
// SubMConv3d / SparseConv3d / SparseInverseConv3d forward and input-gradient, one pair-stationary
// MFMA kernel for the weight gradient.  Replaces spconv's implicit-GEMM kernels behind
// unidet3d/spconv_unet.py:34-72,146-192 and unidet3d/unidet3d.py:96-103.
//
// Forward / dgrad (spconv_gmm_k):
//   * a workgroup (4 wave64) owns a tile of T consecutive DST rows; its fp32 accumulator tile
//     lives in LDS ([T][Cd+4]), so dst is written exactly once (no global atomics, no per-pair
//     feature round trip through HBM: algorithmic traffic N*(Cs+Cd)*4 + pair indices + weights).
//   * the canonical rulebook is the working structure: for offset k the tile's pairs are the
//     contiguous range tile_starts[k][t] .. tile_starts[k][t+1] of the (ascending) scatter list;
//     offsets with no pair in the tile are skipped, no padding work on absent neighbours.
//   * per (offset, channel block): W_k block staged in LDS (double buffered, global loads issued
//     before the MFMA phase and written to LDS after it); each wave takes 16-pair chunks: gathers
//     src rows straight into MFMA A fragments (float4 per lane, K-permuted so one 16-byte load
//     feeds 4 v_mfma_f32_16x16x4_f32), reads B fragments with ds_read_b128 and scatters the
//     16x16 results into the LDS accumulator with ds_add_f32.
//   * fp32 in / fp32 accumulate MFMA (exact fp32, 157 TF peak) -- BASELINE config 2 is fp32.
#include "u3d_common.h"

namespace u3d {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct GmmParams {
    const float* src;
    const float* w;
    const int32_t* gather;
    const int32_t* scatter;
    const int32_t* ts;
    const float* addend;
    float* dst;
    int K;
    int64_t cap;
    int Cs;
    int64_t n_dst;
    int T;
    int64_t n_tiles;
    int ncb;
};

__device__ __forceinline__ void lds_add(float* p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int CB, int CD16>
__device__ __forceinline__ void gmm_gload(float4 (&pf)[(CD16 * 16 * CB * 4 + 255) / 256], const GmmParams& p, int k, int cb, int tid) {
    constexpr int W4 = CD16 * 16 * CB * 4, NPF = (W4 + 255) / 256;
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
        const int idx = (NPF * 256 == W4) ? tid + j * 256 : min(tid + j * 256, W4 - 1);   // clamped: always defined
        const int n = idx / (CB * 4), c4 = idx % (CB * 4);
        pf[j] = *reinterpret_cast<const float4*>(p.w + ((int64_t)n * p.K + k) * p.Cs + cb * (CB * 16) + c4 * 4);
    }
}
template <int CB, int CD16>
__device__ __forceinline__ void gmm_lstore(const float4 (&pf)[(CD16 * 16 * CB * 4 + 255) / 256], float* wb, int tid) {
    constexpr int W4 = CD16 * 16 * CB * 4, NPF = (W4 + 255) / 256, WLD = CB * 16 + 4;
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
        const int idx = tid + j * 256;
        if (NPF * 256 == W4 || idx < W4) {
            const int n = idx / (CB * 4), c4 = idx % (CB * 4);
            *reinterpret_cast<float4*>(wb + n * WLD + c4 * 4) = pf[j];
        }
    }
}

template <int CB, int CD16>
__global__ __launch_bounds__(256) void spconv_gmm_k(GmmParams p) {
    constexpr int CD = CD16 * 16, CBW = CB * 16, WLD = CBW + 4, ALD = CD + 4;
    constexpr int W4 = CD * CB * 4;                  // float4 per staged weight block
    constexpr int NPF = (W4 + 255) / 256;            // float4 per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* acc = smem;
    float* wbuf = smem + (size_t)p.T * ALD;
    int* s_act = reinterpret_cast<int*>(wbuf + 2 * CD * WLD);   // [32] active offsets, [32] = count

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, q = lane >> 4;
    const int64_t t = blockIdx.x;
    const int64_t tile_base = t * p.T;
    const int rows = (int)min((int64_t)p.T, p.n_dst - tile_base);
    const int64_t tsld = p.n_tiles + 1;

    // ---- accumulator init (zeros or the fused residual addend) ----
    for (int idx = tid; idx < rows * (CD / 4); idx += 256) {
        const int r = idx / (CD / 4), c4 = idx % (CD / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.addend) v = *reinterpret_cast<const float4*>(p.addend + (tile_base + r) * CD + c4 * 4);
        *reinterpret_cast<float4*>(acc + r * ALD + c4 * 4) = v;
    }
    // ---- offsets that have at least one pair in this tile ----
    if (wave == 0) {
        bool act = false;
        if (lane < p.K) act = p.ts[lane * tsld + t + 1] > p.ts[lane * tsld + t];
        const unsigned long long m = __ballot(act);
        if (act) s_act[__popcll(m & ((1ull << lane) - 1ull))] = lane;
        if (lane == 0) s_act[32] = __popcll(m);
    }
    __syncthreads();
    const int n_units = s_act[32] * p.ncb;

    float4 pf[NPF];
    if (n_units > 0) {
        gmm_gload<CB, CD16>(pf, p, s_act[0], 0, tid);
        gmm_lstore<CB, CD16>(pf, wbuf, tid);
    }
    for (int u = 0; u < n_units; ++u) {
        __syncthreads();
        {   // unconditional prefetch of the next block (the last iteration re-reads its own: harmless)
            const int un = min(u + 1, n_units - 1);
            gmm_gload<CB, CD16>(pf, p, s_act[un / p.ncb], un % p.ncb, tid);
        }
        {
            const int k = s_act[u / p.ncb], cb = u % p.ncb;
            const float* wb = wbuf + (u & 1) * (CD * WLD);
            const int s = p.ts[k * tsld + t], e = p.ts[k * tsld + t + 1];
            const int32_t* gl = p.gather + (int64_t)k * p.cap;
            const int32_t* sl = p.scatter + (int64_t)k * p.cap;
            const int nchunk = (e - s + 15) >> 4;
            for (int c = wave; c < nchunk; c += 4) {
                const int base = s + c * 16;
                const int g = (base + i16 < e) ? gl[base + i16] : -1;
                float4 a[CB];
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (g >= 0) a[j] = *reinterpret_cast<const float4*>(p.src + (int64_t)g * p.Cs + cb * CBW + j * 16 + q * 4);
                }
                int srow[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int idx = base + q * 4 + r;
                    srow[r] = idx < e ? (int)(sl[idx] - tile_base) : -1;
                }
#pragma unroll
                for (int nb = 0; nb < CD16; nb += 2) {
                    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < CB; ++j) {
                        const float4 b0 = *reinterpret_cast<const float4*>(wb + (nb * 16 + i16) * WLD + j * 16 + q * 4);
                        const float4 b1 = *reinterpret_cast<const float4*>(wb + ((nb + 1) * 16 + i16) * WLD + j * 16 + q * 4);
                        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b0.x, d0, 0, 0, 0);
                        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].x, b1.x, d1, 0, 0, 0);
                        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b0.y, d0, 0, 0, 0);
                        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].y, b1.y, d1, 0, 0, 0);
                        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b0.z, d0, 0, 0, 0);
                        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].z, b1.z, d1, 0, 0, 0);
                        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b0.w, d0, 0, 0, 0);
                        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j].w, b1.w, d1, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (srow[r] >= 0) {
                            lds_add(acc + srow[r] * ALD + nb * 16 + i16, d0[r]);
                            lds_add(acc + srow[r] * ALD + (nb + 1) * 16 + i16, d1[r]);
                        }
                    }
                }
            }
        }
        gmm_lstore<CB, CD16>(pf, wbuf + ((u + 1) & 1) * (CD * WLD), tid);
    }
    __syncthreads();
    for (int idx = tid; idx < rows * (CD / 4); idx += 256) {
        const int r = idx / (CD / 4), c4 = idx % (CD / 4);
        *reinterpret_cast<float4*>(p.dst + (tile_base + r) * CD + c4 * 4) =
            *reinterpret_cast<const float4*>(acc + r * ALD + c4 * 4);
    }
}

// channel blocking: CB 16-column groups of the source per staged weight block
static int pick_cb(int Cs, int Cd) {
    const int cs16 = Cs / 16;
    const int cands[4] = {8, 4, 2, 1};
    for (int c : cands) {
        if (cs16 % c) continue;
        if ((int64_t)Cd * (16 * c + 4) <= 6000) return c;
    }
    return 1;
}
static int64_t gmm_lds_bytes(int CB, int Cd, int T) {
    return ((int64_t)T * (Cd + 4) + 2 * (int64_t)Cd * (CB * 16 + 4)) * 4 + 33 * 4 + 16;
}
static int pick_tile(int Cs, int Cd) {
    const int CB = pick_cb(Cs, Cd);
    const int cands[4] = {256, 128, 64, 32};
    for (int T : cands)
        if (gmm_lds_bytes(CB, Cd, T) <= 80 * 1024) return T;
    return gmm_lds_bytes(CB, Cd, 32) <= 160 * 1024 ? 32 : -1;
}

template <int CB, int CD16>
static int launch_gmm(const GmmParams& p, hipStream_t s) {
    const size_t lds = (size_t)gmm_lds_bytes(CB, CD16 * 16, p.T);
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&spconv_gmm_k<CB, CD16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((spconv_gmm_k<CB, CD16>), dim3((unsigned)p.n_tiles), dim3(256), lds, s, p);
    return check_launch("spconv_gmm");
}

// ------------------------------------------------------------------------------------------
// weight gradient: dW_k[n][c] = sum_p dy[rows_dy[k][p]][n] * x[rows_x[k][p]][c]
// grid (n_split, K); each workgroup reduces a pair range, staging 32..128 gathered rows of x and
// dy in LDS, 16x16x4 fp32 MFMA with the pair index as the reduction dim; wave (wr,wc) owns a
// (Cd/2 x Cs/2) rectangle of dW so operand fragments are reused; one fp32 atomic add per element
// per workgroup at the end.
struct WgParams {
    const float* x;
    const float* dy;
    const int32_t* rows_x;
    const int32_t* rows_dy;
    const int32_t* counts;
    float* dW;
    int K;
    int64_t cap;
    int R;   // pairs per workgroup
};

template <int CS16, int CD16, int PC>
__global__ __launch_bounds__(256) void spconv_wgrad_k(WgParams p) {
    constexpr int CS = CS16 * 16, CD = CD16 * 16, XLD = CS + 4, GLD = CD + 4;
    constexpr int NCI = CS16 >= 2 ? CS16 / 2 : 1, NCO = CD16 / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;
    float* gs = smem + PC * XLD;
    const int k = blockIdx.y;
    const int cnt = p.counts[k];
    const int lo = blockIdx.x * p.R;
    if (lo >= cnt) return;
    const int hi = min(cnt, lo + p.R);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
    const int wr = wave & 1, wc = wave >> 1;
    const bool wave_active = CS16 >= 2 || wc == 0;
    const int32_t* rx = p.rows_x + (int64_t)k * p.cap;
    const int32_t* rg = p.rows_dy + (int64_t)k * p.cap;

    f32x4 d[NCO][NCI];
#pragma unroll
    for (int a = 0; a < NCO; ++a)
#pragma unroll
        for (int b = 0; b < NCI; ++b) d[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int base = lo; base < hi; base += PC) {
        __syncthreads();
        // stage PC gathered rows of x and dy (zeros past the end)
        for (int idx = tid; idx < PC * (CS / 4); idx += 256) {
            const int r = idx / (CS / 4), c4 = idx % (CS / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (base + r < hi) v = *reinterpret_cast<const float4*>(p.x + (int64_t)rx[base + r] * CS + c4 * 4);
            *reinterpret_cast<float4*>(xs + r * XLD + c4 * 4) = v;
        }
        for (int idx = tid; idx < PC * (CD / 4); idx += 256) {
            const int r = idx / (CD / 4), c4 = idx % (CD / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (base + r < hi) v = *reinterpret_cast<const float4*>(p.dy + (int64_t)rg[base + r] * CD + c4 * 4);
            *reinterpret_cast<float4*>(gs + r * GLD + c4 * 4) = v;
        }
        __syncthreads();
        if (wave_active) {
#pragma unroll 2
            for (int kk = 0; kk < PC / 4; ++kk) {
                const int pr = kk * 4 + q;
                float av[NCO], bv[NCI];
#pragma unroll
                for (int a = 0; a < NCO; ++a) av[a] = gs[pr * GLD + (wr * NCO + a) * 16 + i16];
#pragma unroll
                for (int b = 0; b < NCI; ++b) bv[b] = xs[pr * XLD + ((CS16 >= 2 ? wc * NCI : 0) + b) * 16 + i16];
#pragma unroll
                for (int a = 0; a < NCO; ++a)
#pragma unroll
                    for (int b = 0; b < NCI; ++b) d[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], d[a][b], 0, 0, 0);
            }
        }
    }
    if (wave_active) {
#pragma unroll
        for (int a = 0; a < NCO; ++a)
#pragma unroll
            for (int b = 0; b < NCI; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = (wr * NCO + a) * 16 + q * 4 + r;
                    const int c = ((CS16 >= 2 ? wc * NCI : 0) + b) * 16 + i16;
                    atomicAdd(p.dW + ((int64_t)n * p.K + k) * CS + c, d[a][b][r]);
                }
    }
}

template <int CS16, int CD16>
static int launch_wgrad(const WgParams& p0, int64_t n_rows_hint, hipStream_t s) {
    constexpr int total = (CS16 + CD16) * 16;
    constexpr int PC = total <= 64 ? 128 : (total <= 192 ? 64 : 32);
    WgParams p = p0;
    int64_t R = ceil_div(p.cap, (int64_t)(1024 / p.K > 0 ? 1024 / p.K : 1));
    R = ceil_div(R, PC) * PC;
    if (R < 2 * PC) R = 2 * PC;
    p.R = (int)R;
    const int64_t nsplit = ceil_div(p.cap, R);
    const size_t lds = (size_t)PC * ((CS16 + CD16) * 16 + 8) * 4;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&spconv_wgrad_k<CS16, CD16, PC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL((spconv_wgrad_k<CS16, CD16, PC>), dim3((unsigned)nsplit, p.K), dim3(256), lds, s, p);
    return check_launch("spconv_wgrad");
}

__global__ void weight_transpose_k(const float* __restrict__ w, float* __restrict__ wt, int Cd, int K, int Cs) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)Cd * K * Cs;
    if (idx >= total) return;
    // idx enumerates wt[(c*K + k)*Cd + n]
    const int n = (int)(idx % Cd);
    const int k = (int)((idx / Cd) % K);
    const int c = (int)(idx / ((int64_t)Cd * K));
    wt[idx] = w[((int64_t)n * K + k) * Cs + c];
}

}  // namespace u3d

using namespace u3d;

extern "C" {

int u3d_spconv_tile_rows(int Cs, int Cd) {
    if (Cs % 16 || Cd % 32 || Cs <= 0 || Cd <= 0 || Cd > 256 || Cs > 256) return U3D_EUNSUPPORTED;
    return pick_tile(Cs, Cd);
}

int u3d_spconv_gmm(const float* src, const float* w_rows, const int32_t* gather, const int32_t* scatter,
                   const int32_t* tile_starts, int K, int64_t cap, int Cs, int Cd, int64_t n_dst, int tile_rows,
                   const float* addend, float* dst, double flops_hint, u3d_stream_t stream) {
    if (!src || !w_rows || !gather || !scatter || !tile_starts || !dst || K <= 0 || K > 32 || n_dst <= 0) return U3D_EINVAL;
    if (Cs % 16 || Cd % 32 || tile_rows != pick_tile(Cs, Cd)) {
        set_error("spconv_gmm: unsupported Cs=%d Cd=%d tile=%d", Cs, Cd, tile_rows);
        return U3D_EUNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_CONV_FWD, s, flops_hint);
    GmmParams p;
    p.src = src; p.w = w_rows; p.gather = gather; p.scatter = scatter; p.ts = tile_starts; p.addend = addend; p.dst = dst;
    p.K = K; p.cap = cap; p.Cs = Cs; p.n_dst = n_dst; p.T = tile_rows; p.n_tiles = ceil_div(n_dst, tile_rows);
    const int CB = pick_cb(Cs, Cd);
    p.ncb = Cs / (CB * 16);
    const int cd16 = Cd / 16;
#define U3D_GMM_CASE(cb, cd) if (CB == cb && cd16 == cd) return launch_gmm<cb, cd>(p, s);
    U3D_GMM_CASE(1, 2) U3D_GMM_CASE(2, 2) U3D_GMM_CASE(4, 2) U3D_GMM_CASE(8, 2)
    U3D_GMM_CASE(1, 4) U3D_GMM_CASE(2, 4) U3D_GMM_CASE(4, 4)
    U3D_GMM_CASE(1, 6) U3D_GMM_CASE(2, 6)
    U3D_GMM_CASE(1, 8) U3D_GMM_CASE(2, 8)
    U3D_GMM_CASE(1, 10) U3D_GMM_CASE(2, 10)
    U3D_GMM_CASE(1, 12) U3D_GMM_CASE(1, 16)
#undef U3D_GMM_CASE
    set_error("spconv_gmm: no instantiation for CB=%d Cd=%d", CB, Cd);
    return U3D_EUNSUPPORTED;
}

int u3d_spconv_wgrad(const float* x, const float* dy, const int32_t* rows_x, const int32_t* rows_dy,
                     const int32_t* counts, int K, int64_t cap, int Cs, int Cd, float* dW, double flops_hint,
                     u3d_stream_t stream) {
    if (!x || !dy || !rows_x || !rows_dy || !counts || !dW || K <= 0 || cap <= 0) return U3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_CONV_WGRAD, s, flops_hint);
    WgParams p;
    p.x = x; p.dy = dy; p.rows_x = rows_x; p.rows_dy = rows_dy; p.counts = counts; p.dW = dW; p.K = K; p.cap = cap; p.R = 0;
    const int cs16 = Cs / 16, cd16 = Cd / 16;
    if (Cs % 16 || Cd % 32) return U3D_EUNSUPPORTED;
#define U3D_WG_CASE(cs, cd) if (cs16 == cs && cd16 == cd) return launch_wgrad<cs, cd>(p, cap, s);
    U3D_WG_CASE(1, 2) U3D_WG_CASE(2, 2) U3D_WG_CASE(4, 2) U3D_WG_CASE(4, 4) U3D_WG_CASE(8, 4)
    U3D_WG_CASE(6, 6) U3D_WG_CASE(12, 6) U3D_WG_CASE(8, 8) U3D_WG_CASE(16, 8) U3D_WG_CASE(10, 10)
    U3D_WG_CASE(2, 4) U3D_WG_CASE(4, 6) U3D_WG_CASE(6, 8) U3D_WG_CASE(8, 10)
    U3D_WG_CASE(6, 4) U3D_WG_CASE(8, 6) U3D_WG_CASE(10, 8)
#undef U3D_WG_CASE
    set_error("spconv_wgrad: no instantiation for Cs=%d Cd=%d", Cs, Cd);
    return U3D_EUNSUPPORTED;
}

int u3d_weight_transpose(const float* w, float* wt, int Cd, int K, int Cs, u3d_stream_t stream) {
    if (!w || !wt || Cd <= 0 || K <= 0 || Cs <= 0) return U3D_EINVAL;
    const int64_t total = (int64_t)Cd * K * Cs;
    hipLaunchKernelGGL(weight_transpose_k, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wt, Cd, K, Cs);
    return check_launch("weight_transpose");
}

}  // extern "C"
