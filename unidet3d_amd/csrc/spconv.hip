// K4-K8 sparse convolutions on gfx950: one output-stationary gather-MFMA-scatter kernel for
// SubMConv3d / SparseConv3d / SparseInverseConv3d forward and input-gradient, one pair-stationary
// MFMA kernel for the weight gradient.  Replaces spconv's implicit-GEMM kernels behind
// unidet3d/spconv_unet.py:34-72,146-192 and unidet3d/unidet3d.py:96-103.
//
// Forward / dgrad (spconv_gmm_k), wave-independent design (no workgroup barriers at all):
//   * every wave64 owns R (32 or 64) consecutive DST rows x a 32-column slice of the output and,
//     optionally, one of G groups of kernel offsets; its fp32 accumulator tile [R][36] is private
//     LDS, so dst is written exactly once per (row, column) -- no global atomics, no per-pair
//     feature round trip through HBM (algorithmic traffic N*(Cs+Cd)*4 + pair indices + weights).
//   * the canonical rulebook is the working structure: for offset k the wave's pairs are the
//     contiguous range tile_starts[k][t] .. tile_starts[k][t+1] of the ascending scatter list;
//     offsets without a pair in the tile cost two scalar loads, absent neighbours cost nothing.
//   * per offset the wave takes 16-pair chunks two at a time: gathers the src rows straight into
//     MFMA A fragments (one float4 per lane, K-permuted so a 16-byte load feeds four
//     v_mfma_f32_16x16x4_f32), reads the B fragments W_k[n][c..c+3] with one float4 per lane from
//     L2 (weights are [n][k][c], so the fragment is contiguous) shared by both chunks, accumulates
//     over all source channels in registers and adds the 16x32 results into LDS with ds_add_f32.
//   * 4 independent waves per workgroup, 16-18 KB LDS per wave-tile -> up to 16 waves per CU hide the
//     gather latency; deep U-Net levels (a few thousand rows) get their parallelism from column
//     slices and offset groups (partials summed by a small deterministic reduce kernel).
//   * fp32 in / fp32 accumulate MFMA (exact fp32, 157 TF peak) -- BASELINE config 2 is fp32.
#include <stdlib.h>

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
    float* out;          // dst, or the partial buffer [G][n_dst][Cd] when G > 1
    int K;
    int64_t cap;
    int Cs, Cd;
    int64_t n_dst;
    int64_t n_sub;
    int n_slices;
    int G;
    int kper;
};

__device__ __forceinline__ void lds_add(float* p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

constexpr int GMM_CDS = 32;            // output columns per wave
constexpr int GMM_ALD = GMM_CDS + 4;   // padded accumulator row

// raw rulebook indices of one work item (two 16-pair chunks of one offset), loaded one item ahead
struct GmmIdx {
    int g, s;      // lane l < 32 holds gather / scatter row of pair (base + l) of the work item
};

// Two coalesced 128-byte loads per item (lanes 0..31 = the item's 32 pairs; indices clamped into the range so the
// load is unconditional); gmm_chunks distributes them to the MFMA fragment layout with shuffles.
__device__ __forceinline__ void gmm_load_idx(GmmIdx& ix, const GmmParams& p, int k, int base, int e, int lane) {
    const int pi = min(base + (lane & 31), e - 1);
    ix.g = p.gather[(int64_t)k * p.cap + pi];
    ix.s = p.scatter[(int64_t)k * p.cap + pi];
}

// one or two 16-pair chunks of offset k: gather -> MFMA over all source channels -> scatter-add into LDS
template <int CS16, bool TWO, int TRASH>
__device__ __forceinline__ void gmm_chunks(const GmmParams& p, GmmIdx ix, const float* __restrict__ wk, int base, int e,
                                           int64_t row0, float* acc, int i16, int q) {
    constexpr int JB = CS16 <= 8 ? CS16 : CS16 / 2;     // 16-channel groups held in registers at a time
    constexpr int NJB = CS16 / JB;
    // lanes past the end of the range hold a clamped (valid) index: their rows are computed but only ever
    // added into the scratch row, so nothing here waits on a branch.
    const int g0 = __shfl(ix.g, i16, 64), g1 = TWO ? __shfl(ix.g, 16 + i16, 64) : 0;
    int srow0[4], srow1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i0 = base + q * 4 + r, i1 = i0 + 16;
        const int v0 = __shfl(ix.s, q * 4 + r, 64);
        const int v1 = TWO ? __shfl(ix.s, 16 + q * 4 + r, 64) : 0;
        srow0[r] = i0 < e ? (int)(v0 - row0) : TRASH;        // lanes past the end add into a scratch row
        srow1[r] = (TWO && i1 < e) ? (int)(v1 - row0) : TRASH;
    }
    f32x4 d0[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 d1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const float* a0p = p.src + (int64_t)g0 * p.Cs + q * 4;
    const float* a1p = p.src + (int64_t)g1 * p.Cs + q * 4;
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) {
        float4 a0[JB], a1[JB];
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            a0[j] = *reinterpret_cast<const float4*>(a0p + (jb * JB + j) * 16);
            if (TWO) a1[j] = *reinterpret_cast<const float4*>(a1p + (jb * JB + j) * 16);
        }
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const float4 b0 = *reinterpret_cast<const float4*>(wk + ((jb * JB + j) * 2 + 0) * 256);    // 1 KB contiguous per wave
            const float4 b1 = *reinterpret_cast<const float4*>(wk + ((jb * JB + j) * 2 + 1) * 256);
            // independent accumulator chains interleaved (16x16x4 f32: 32-cycle issue, 40-cycle dependent latency)
#define U3D_STEP(c)                                                                          \
    d0[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j].c, b0.c, d0[0], 0, 0, 0);             \
    d0[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j].c, b1.c, d0[1], 0, 0, 0);             \
    if (TWO) {                                                                               \
        d1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j].c, b0.c, d1[0], 0, 0, 0);         \
        d1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j].c, b1.c, d1[1], 0, 0, 0);         \
    }
            U3D_STEP(x) U3D_STEP(y) U3D_STEP(z) U3D_STEP(w)
#undef U3D_STEP
        }
    }
    // scatter: the accumulator tile is private to this wave and, inside one offset, every destination row
    // occurs at most once -> a plain LDS read-modify-write is exact (ds_add_f32 atomics measured ~10x slower:
    // SQ_WAIT_INST_LDS was 83 % of all wave cycles with them).
    float o0[2][4], o1[2][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o0[nb][r] = acc[srow0[r] * GMM_ALD + nb * 16 + i16];
            if (TWO) o1[nb][r] = acc[srow1[r] * GMM_ALD + nb * 16 + i16];
        }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[srow0[r] * GMM_ALD + nb * 16 + i16] = o0[nb][r] + d0[nb][r];
            if (TWO) acc[srow1[r] * GMM_ALD + nb * 16 + i16] = o1[nb][r] + d1[nb][r];
        }
}

template <int CS16, int R>
__global__ __launch_bounds__(256) void spconv_gmm_k(GmmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, q = lane >> 4;
    float* acc = smem + wave * ((R + 1) * GMM_ALD);      // R rows + one scratch row

    const int64_t wid = xcd_swizzle(blockIdx.x, gridDim.x) * 4 + wave;     // neighbouring row tiles share an XCD / L2
    const int per_sub = p.n_slices * p.G;
    const int64_t sub = wid / per_sub;
    if (sub >= p.n_sub) return;                      // wave-uniform; there are no barriers in this kernel
    const int rem = (int)(wid % per_sub);
    const int slice = rem / p.G, g = rem % p.G;
    const int n0 = slice * GMM_CDS;
    const int64_t row0 = sub * R;
    const int rows = (int)min((int64_t)R, p.n_dst - row0);
    const int64_t tsld = p.n_sub + 1;

    // ---- accumulator init: zeros, or the fused residual addend (single offset group only) ----
    for (int idx = lane; idx < rows * (GMM_CDS / 4); idx += 64) {
        const int r = idx >> 3, c4 = idx & 7;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.addend && p.G == 1) v = *reinterpret_cast<const float4*>(p.addend + (row0 + r) * p.Cd + n0 + c4 * 4);
        *reinterpret_cast<float4*>(acc + r * GMM_ALD + c4 * 4) = v;
    }

    const int k_lo = g * p.kper, k_hi = min(p.K, k_lo + p.kper);
    // all (start, end) ranges of this wave's offsets in one round trip: lane k holds offset k's range
    int ts_s = 0, ts_e = 0;
    if (lane < p.K) {
        ts_s = p.ts[lane * tsld + sub];
        ts_e = p.ts[lane * tsld + sub + 1];
    }
    // work items = (offset k, 32-pair window); the raw indices of item i+1 are loaded while item i computes
    auto range_of = [&](int k, int& s_, int& e_) {
        s_ = __builtin_amdgcn_readfirstlane(__shfl(ts_s, k, 64));      // wave-uniform -> scalar control flow
        e_ = __builtin_amdgcn_readfirstlane(__shfl(ts_e, k, 64));
    };
    int k = k_lo, base = 0, e = 0;
    for (; k < k_hi; ++k) {                     // first non-empty offset
        range_of(k, base, e);
        if (base < e) break;
    }
    GmmIdx cur;
    if (k < k_hi) gmm_load_idx(cur, p, k, base, e, lane);
    while (k < k_hi) {
        // ---- locate the next item and start its index loads ----
        int nk = k, nbase = base + 32, ne = e;
        if (nbase >= e) {
            for (nk = k + 1; nk < k_hi; ++nk) {
                range_of(nk, nbase, ne);
                if (nbase < ne) break;
            }
        }
        GmmIdx nxt = cur;
        if (nk < k_hi) gmm_load_idx(nxt, p, nk, nbase, ne, lane);
        // ---- compute the current item ----
        const float* wk = p.w + ((int64_t)slice * p.K + k) * (CS16 * 512) + lane * 4;   // packed fragments of (slice, k)
        if (base + 16 < e) gmm_chunks<CS16, true, R>(p, cur, wk, base, e, row0, acc, i16, q);
        else gmm_chunks<CS16, false, R>(p, cur, wk, base, e, row0, acc, i16, q);
        cur = nxt; k = nk; base = nbase; e = ne;
    }
    float* out = p.out + (p.G > 1 ? (int64_t)g * p.n_dst * p.Cd : 0);
    for (int idx = lane; idx < rows * (GMM_CDS / 4); idx += 64) {
        const int r = idx >> 3, c4 = idx & 7;
        *reinterpret_cast<float4*>(out + (row0 + r) * p.Cd + n0 + c4 * 4) = *reinterpret_cast<const float4*>(acc + r * GMM_ALD + c4 * 4);
    }
}

// dst = sum_g partial[g] (+ addend), fixed summation order
__global__ __launch_bounds__(256) void gmm_reduce_k(const float* __restrict__ partial, int G, int64_t n4, const float* __restrict__ addend,
                                                    float* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = addend ? reinterpret_cast<const float4*>(addend)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 0; g < G; ++g) {
            const float4 t = reinterpret_cast<const float4*>(partial)[(int64_t)g * n4 + i];
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        reinterpret_cast<float4*>(dst)[i] = v;
    }
}

// rows per wave-tile and offset groups: aim at >= 2 waves per SIMD on 256 CUs x 4 SIMDs
static void plan_gmm(int Cs, int Cd, int K, int64_t n_dst, int* R, int* G) {
    const int slices = Cd / GMM_CDS;
    const int64_t want = 2048;
    int r = 64, g = 1;
    if (ceil_div(n_dst, 64) * slices < want) r = 32;
    const int64_t waves = ceil_div(n_dst, r) * slices;
    if (waves < want && K >= 27) g = waves * 3 >= want ? 3 : 9;
    if (const char* e = getenv("U3D_GMM_R")) r = atoi(e) == 32 ? 32 : 64;      // experiment knob (tools/prof_conv.py)
    *R = r;
    *G = g;
}

template <int CS16, int R>
static int launch_gmm(const GmmParams& p, hipStream_t s) {
    const size_t lds = (size_t)4 * (R + 1) * GMM_ALD * sizeof(float);
    const int64_t waves = p.n_sub * p.n_slices * p.G;
    hipLaunchKernelGGL((spconv_gmm_k<CS16, R>), dim3((unsigned)ceil_div(waves, 4)), dim3(256), lds, s, p);
    return check_launch("spconv_gmm");
}

// ------------------------------------------------------------------------------------------
// weight gradient: dW_k[n][c] = sum_p dy[rows_dy[k][p]][n] * x[rows_x[k][p]][c]
// Pair-stationary, barrier-free: a wave walks a contiguous range of offset k's pair list and keeps its
// (sub-)block of dW_k in MFMA accumulators.  The 16x16x4 fp32 MFMA takes the PAIR index as its
// reduction dim (4 pairs per instruction) and the channels as M / N.  Channel blocks are STRIDED
// (block s = channels {NG*i + s}): lane i16 then needs NG consecutive floats of its pair's row, so one
// load instruction fetches four complete, contiguous rows (16 lanes x NG*4 bytes each) -- a fraction of
// the cache-line look-ups of a fragment-shaped gather and no LDS staging at all.  Large channel counts are
// split over the 4 waves of a workgroup (sub-blocks of <= 32 accumulators); for small ones the 4 waves
// take 4 ranges and pre-reduce in LDS so that a workgroup issues one fp32 atomic per dW element.
struct WgParams {
    const float* x;
    const float* dy;
    const int32_t* rows_x;
    const int32_t* rows_dy;
    const int32_t* ts;        // tile_starts [K][n_tiles+1] over the dy-side rows: range t of offset k = pairs of row tile t
    float* partial;           // [K][n_tiles][Cd*Cs] per-range blocks in accumulator-register order
    int K;
    int64_t cap;
    int n_tiles;
};

template <int N>
__device__ __forceinline__ void load_row_part(float (&v)[N], const float* __restrict__ p) {
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (int s = 0; s < N; s += 4) {
            const float4 t = *reinterpret_cast<const float4*>(p + s);
            v[s] = t.x; v[s + 1] = t.y; v[s + 2] = t.z; v[s + 3] = t.w;
        }
    } else if constexpr (N % 2 == 0) {
#pragma unroll
        for (int s = 0; s < N; s += 2) {
            const float2 t = *reinterpret_cast<const float2*>(p + s);
            v[s] = t.x; v[s + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int s = 0; s < N; ++s) v[s] = p[s];
    }
}

// NG = Cd/16, NX = Cs/16 floats per lane per row; SG x SX waves share one pair range
template <int NG, int NX, int SG, int SX>
__global__ __launch_bounds__(256) void spconv_wgrad_k(WgParams p) {
    constexpr int NGW = NG / SG, NXW = NX / SX, SPLIT = SG * SX, RPW = 4 / SPLIT;   // RPW ranges per workgroup
    constexpr int CD = NG * 16, CS = NX * 16;
    // A range is the set of pairs of offset k whose dy row lies in row tile t (tile_starts), so the 27 offsets of
    // one tile read the SAME dy rows and neighbouring x rows.  Workgroup b runs on XCD b % 8 (private 4 MB L2):
    // XCD x takes tile groups x, x+8, x+16, ... and walks ALL offsets of a group back to back, so those rows are
    // fetched once and re-read from the XCD's own L2 (PMC with pair-index ranges: TCC hit rate 11 %, 0.95 GB
    // fetched per L1 launch for 125 MB of algorithmic traffic).
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int k = j % p.K;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
    const int sub = wave % SPLIT, wa = sub % SG, wb = sub / SG;
    const int range = ((j / p.K) * 8 + xcd) * RPW + wave / SPLIT;
    if (range >= p.n_tiles) return;                      // wave-uniform; the kernel has no barrier
    const int lo = p.ts[(int64_t)k * (p.n_tiles + 1) + range];
    const int hi = p.ts[(int64_t)k * (p.n_tiles + 1) + range + 1];
    const int32_t* rx = p.rows_x + (int64_t)k * p.cap;
    const int32_t* rg = p.rows_dy + (int64_t)k * p.cap;
    const float* gbase = p.dy + NG * i16 + wa * NGW;
    const float* xbase = p.x + NX * i16 + wb * NXW;

    f32x4 acc[NGW][NXW];
#pragma unroll
    for (int a = 0; a < NGW; ++a)
#pragma unroll
        for (int b = 0; b < NXW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int base = lo; base < hi; base += 16) {        // 4 MFMA K-steps (16 pairs) per trip, all loads up front
        int io[4], ix[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pi = min(base + 4 * u + q, hi - 1);
            io[u] = rg[pi];
            ix[u] = rx[pi];
        }
        float gv[4][NGW], xv[4][NXW];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            load_row_part<NGW>(gv[u], gbase + (int64_t)io[u] * CD);
            load_row_part<NXW>(xv[u], xbase + (int64_t)ix[u] * CS);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = base + 4 * u + q < hi;        // pairs past the end contribute exact zeros
#pragma unroll
            for (int a = 0; a < NGW; ++a) {
                const float ga = ok ? gv[u][a] : 0.f;
#pragma unroll
                for (int b = 0; b < NXW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga, xv[u][b], acc[a][b], 0, 0, 0);
            }
        }
    }
    // partial block of this range in register order [sub][(a*NXW + b)*4 + r][lane]: 256-byte coalesced stores,
    // no atomics; wgrad_reduce_k maps it back to dW[co][k][ci] and sums the ranges in a fixed order.
    float* out = p.partial + ((int64_t)k * p.n_tiles + range) * (CD * CS) + (int64_t)sub * (NGW * NXW * 256) + lane;
#pragma unroll
    for (int a = 0; a < NGW; ++a)
#pragma unroll
        for (int b = 0; b < NXW; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[((a * NXW + b) * 4 + r) * 64] = acc[a][b][r];
}

// dW[co][k][ci] = sum over the row tiles of offset k (fixed order -> deterministic; overwrites dW)
__global__ __launch_bounds__(256) void wgrad_reduce_k(const float* __restrict__ partial, int n_tiles, int K, int NG, int NX, int SG, int SX,
                                                      float* __restrict__ dW) {
    const int CS = NX * 16, E = NG * NX * 256;
    const int k = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= E) return;
    const float* src = partial + (int64_t)k * n_tiles * E + idx;
    // four independent partial sums keep loads in flight; the combination order is fixed (deterministic)
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int r = 0;
    for (; r + 4 <= n_tiles; r += 4) {
        v0 += src[(int64_t)r * E];
        v1 += src[(int64_t)(r + 1) * E];
        v2 += src[(int64_t)(r + 2) * E];
        v3 += src[(int64_t)(r + 3) * E];
    }
    for (; r < n_tiles; ++r) v0 += src[(int64_t)r * E];
    const float v = (v0 + v1) + (v2 + v3);
    const int NGW = NG / SG, NXW = NX / SX, per_sub = NGW * NXW * 256;
    const int sub = idx / per_sub, rem = idx % per_sub;
    const int e = rem >> 6, lane = rem & 63, i16 = lane & 15, q = lane >> 4;
    const int rr = e & 3, b = (e >> 2) % NXW, a = (e >> 2) / NXW;
    const int wa = sub % SG, wb = sub / SG;
    const int co = NG * (4 * q + rr) + wa * NGW + a, ci = NX * i16 + wb * NXW + b;
    dW[((int64_t)co * K + k) * CS + ci] = v;
}

// rows per tile: as many tiles as a 64 MB partial buffer, a cap of 512 and >= 64 rows per tile allow
static int plan_wgrad_rows(int K, int64_t n_rows, int Cs, int Cd) {
    int64_t nt = (int64_t)(64 << 20) / ((int64_t)K * Cs * Cd * 4);
    nt = nt > 512 ? 512 : (nt < 1 ? 1 : nt);
    const int64_t by_len = ceil_div(n_rows, 64);
    if (nt > by_len) nt = by_len;
    return (int)ceil_div(n_rows, nt);
}

template <int NX, int NG>     // (Cs/16, Cd/16) as the dispatch macro passes them
static int launch_wgrad(const WgParams& p0, float* dW, hipStream_t s) {
    // split over the 4 waves until a wave's sub-block is <= 32 accumulators (128 VGPRs)
    constexpr int SG = (NG * NX > 32 && NG % 2 == 0 && NG >= NX) ? 2 : ((NG * NX > 64 && NG % 2 == 0) ? 2 : 1);
    constexpr int SX = ((NG / SG) * NX > 32 && NX % 2 == 0) ? 2 : 1;
    constexpr int SPLIT = SG * SX, RPW = 4 / SPLIT;
    static_assert(SPLIT == 1 || SPLIT == 2 || SPLIT == 4, "wave split");
    const WgParams& p = p0;
    const int64_t groups = ceil_div(ceil_div(p.n_tiles, RPW), 8) * 8;
    hipLaunchKernelGGL((spconv_wgrad_k<NG, NX, SG, SX>), dim3((unsigned)(groups * p.K)), dim3(256), 0, s, p);
    hipLaunchKernelGGL(wgrad_reduce_k, dim3(NG * NX, p.K), dim3(256), 0, s, (const float*)p.partial, p.n_tiles, p.K, NG, NX, SG, SX, dW);
    return check_launch("spconv_wgrad");
}

// wp[(((slice*K + k)*CS16 + j)*2 + nb)*256 + lane*4 + t] = W(n = slice*32 + nb*16 + (lane&15), k, c = j*16 + (lane>>4)*4 + t)
// i.e. the B fragments of spconv_gmm_k in the order the kernel reads them (one contiguous 1 KB block per wave load).
// transposed = 0: W(n,k,c) = w[(n*K + k)*Cs + c]  (forward, w = [Cd][K][Cs]);
// transposed = 1: W(n,k,c) = w[(c*K + k)*Cd + n]  (input gradient: w = [Cs][K][Cd] is the forward weight).
__global__ __launch_bounds__(256) void weight_pack_k(const float* __restrict__ w, float* __restrict__ wp, int Cd, int K, int Cs, int transposed) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one float4 per thread
    const int cs16 = Cs / 16;
    const int64_t total = (int64_t)(Cd / 32) * K * cs16 * 2 * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    int64_t t = idx >> 6;
    const int nb = (int)(t & 1); t >>= 1;
    const int j = (int)(t % cs16); t /= cs16;
    const int k = (int)(t % K);
    const int slice = (int)(t / K);
    const int n = slice * 32 + nb * 16 + (lane & 15), c = j * 16 + (lane >> 4) * 4;
    float4 v;
    if (!transposed) {
        v = *reinterpret_cast<const float4*>(w + ((int64_t)n * K + k) * Cs + c);
    } else {
        v.x = w[((int64_t)(c + 0) * K + k) * Cd + n];
        v.y = w[((int64_t)(c + 1) * K + k) * Cd + n];
        v.z = w[((int64_t)(c + 2) * K + k) * Cd + n];
        v.w = w[((int64_t)(c + 3) * K + k) * Cd + n];
    }
    reinterpret_cast<float4*>(wp)[idx] = v;
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

int u3d_spconv_plan(int Cs, int Cd, int K, int64_t n_dst, int* tile_rows, int* k_groups) {
    if (Cs % 16 || Cd % 32 || Cs <= 0 || Cd <= 0 || Cd > 256 || Cs > 256 || K <= 0 || K > 32 || n_dst <= 0 || !tile_rows || !k_groups)
        return U3D_EUNSUPPORTED;
    const int cs16 = Cs / 16;
    if (!(cs16 == 1 || cs16 == 2 || cs16 == 4 || cs16 == 6 || cs16 == 8 || cs16 == 10 || cs16 == 12 || cs16 == 16)) return U3D_EUNSUPPORTED;
    plan_gmm(Cs, Cd, K, n_dst, tile_rows, k_groups);
    return U3D_OK;
}

int u3d_spconv_gmm(const float* src, const float* w_rows, const int32_t* gather, const int32_t* scatter,
                   const int32_t* tile_starts, int K, int64_t cap, int Cs, int Cd, int64_t n_dst, int tile_rows,
                   int k_groups, const float* addend, float* dst, void* ws, double flops_hint, u3d_stream_t stream) {
    if (!src || !w_rows || !gather || !scatter || !tile_starts || !dst || K <= 0 || K > 32 || n_dst <= 0) return U3D_EINVAL;
    int R = 0, G = 0;
    if (u3d_spconv_plan(Cs, Cd, K, n_dst, &R, &G) != U3D_OK || R != tile_rows || G != k_groups || (G > 1 && !ws)) {
        set_error("spconv_gmm: unsupported Cs=%d Cd=%d or plan mismatch (tile %d/%d groups %d/%d)", Cs, Cd, tile_rows, R, k_groups, G);
        return U3D_EUNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_CONV_FWD, s, flops_hint);
    GmmParams p;
    p.src = src; p.w = w_rows; p.gather = gather; p.scatter = scatter; p.ts = tile_starts; p.addend = addend;
    p.out = G > 1 ? (float*)ws : dst;
    p.K = K; p.cap = cap; p.Cs = Cs; p.Cd = Cd; p.n_dst = n_dst; p.n_sub = ceil_div(n_dst, R);
    p.n_slices = Cd / GMM_CDS; p.G = G; p.kper = (int)ceil_div(K, G);
    const int cs16 = Cs / 16;
    int rc = U3D_EUNSUPPORTED;
#define U3D_GMM_CASE(cs) if (cs16 == cs) rc = (R == 64) ? launch_gmm<cs, 64>(p, s) : launch_gmm<cs, 32>(p, s);
    U3D_GMM_CASE(1) U3D_GMM_CASE(2) U3D_GMM_CASE(4) U3D_GMM_CASE(6) U3D_GMM_CASE(8) U3D_GMM_CASE(10) U3D_GMM_CASE(12) U3D_GMM_CASE(16)
#undef U3D_GMM_CASE
    if (rc != U3D_OK) return rc;
    if (G > 1) {
        const int64_t n4 = n_dst * Cd / 4;
        int64_t grid = ceil_div(n4, 256);
        grid = grid > 2048 ? 2048 : grid;
        hipLaunchKernelGGL(gmm_reduce_k, dim3((unsigned)grid), dim3(256), 0, s, (const float*)ws, G, n4, addend, dst);
        rc = check_launch("gmm_reduce");
    }
    return rc;
}

int u3d_spconv_wgrad_tile_rows(int K, int64_t n_rows_dy, int Cs, int Cd) {
    if (K <= 0 || n_rows_dy <= 0 || Cs <= 0 || Cd <= 0) return U3D_EINVAL;
    return plan_wgrad_rows(K, n_rows_dy, Cs, Cd);
}

int64_t u3d_spconv_wgrad_ws_bytes(int K, int64_t n_rows_dy, int Cs, int Cd) {
    if (K <= 0 || n_rows_dy <= 0 || Cs <= 0 || Cd <= 0) return 0;
    const int T = plan_wgrad_rows(K, n_rows_dy, Cs, Cd);
    return (int64_t)K * ceil_div(n_rows_dy, T) * Cs * Cd * 4 + 256;
}

int u3d_spconv_wgrad(const float* x, const float* dy, const int32_t* rows_x, const int32_t* rows_dy,
                     const int32_t* tile_starts, int K, int64_t cap, int64_t n_rows_dy, int tile_rows, int Cs, int Cd,
                     float* dW, void* ws, double flops_hint, u3d_stream_t stream) {
    if (!x || !dy || !rows_x || !rows_dy || !tile_starts || !dW || !ws || K <= 0 || cap <= 0 || n_rows_dy <= 0) return U3D_EINVAL;
    if (tile_rows != plan_wgrad_rows(K, n_rows_dy, Cs, Cd)) {
        set_error("spconv_wgrad: tile_rows %d does not match the plan", tile_rows);
        return U3D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_CONV_WGRAD, s, flops_hint);
    WgParams p;
    p.x = x; p.dy = dy; p.rows_x = rows_x; p.rows_dy = rows_dy; p.ts = tile_starts; p.partial = (float*)ws; p.K = K; p.cap = cap;
    p.n_tiles = (int)ceil_div(n_rows_dy, tile_rows);
    const int cs16 = Cs / 16, cd16 = Cd / 16;
    if (Cs % 16 || Cd % 32) return U3D_EUNSUPPORTED;
#define U3D_WG_CASE(cs, cd) if (cs16 == cs && cd16 == cd) return launch_wgrad<cs, cd>(p, dW, s);
    U3D_WG_CASE(1, 2) U3D_WG_CASE(2, 2) U3D_WG_CASE(4, 2) U3D_WG_CASE(4, 4) U3D_WG_CASE(8, 4)
    U3D_WG_CASE(6, 6) U3D_WG_CASE(12, 6) U3D_WG_CASE(8, 8) U3D_WG_CASE(16, 8) U3D_WG_CASE(10, 10)
    U3D_WG_CASE(2, 4) U3D_WG_CASE(4, 6) U3D_WG_CASE(6, 8) U3D_WG_CASE(8, 10)
    U3D_WG_CASE(6, 4) U3D_WG_CASE(8, 6) U3D_WG_CASE(10, 8)
#undef U3D_WG_CASE
    set_error("spconv_wgrad: no instantiation for Cs=%d Cd=%d", Cs, Cd);
    return U3D_EUNSUPPORTED;
}

int u3d_weight_pack(const float* w, float* wp, int Cd, int K, int Cs, int transposed, u3d_stream_t stream) {
    if (!w || !wp || Cd <= 0 || K <= 0 || Cs <= 0 || Cd % 32 || Cs % 16) return U3D_EINVAL;
    const int64_t total4 = (int64_t)Cd * K * Cs / 4;
    hipLaunchKernelGGL(weight_pack_k, dim3((unsigned)ceil_div(total4, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, Cd, K, Cs, transposed);
    return check_launch("weight_pack");
}

int u3d_weight_transpose(const float* w, float* wt, int Cd, int K, int Cs, u3d_stream_t stream) {
    if (!w || !wt || Cd <= 0 || K <= 0 || Cs <= 0) return U3D_EINVAL;
    const int64_t total = (int64_t)Cd * K * Cs;
    hipLaunchKernelGGL(weight_transpose_k, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wt, Cd, K, Cs);
    return check_launch("weight_transpose");
}

}  // extern "C"
