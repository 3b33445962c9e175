// Sparse-convolution weight gradient from bf16 ROWS (BASELINE configs[2] with bf16 shadows of the gathered activations in HBM,
// bn.hip store_shadow):   dW_k[co][ci] = sum_p dy[rows_dy[k][p]][co] * x[rows_x[k][p]][ci]
// The reduction runs over PAIRS, so both MFMA operands want eight consecutive pairs of one channel per lane -- a column of the
// gathered [pair][channel] tiles.  gfx950's LDS transpose read does exactly that turn: the wave gathers whole bf16 rows (a lane
// loads 16 bytes, 64 / (C / 8) complete rows per instruction), writes them row-major into a private LDS tile and reads the
// fragments back with ds_read_b64_tr_b16 (a 16-lane group hands in a [4 pairs][16 channels] block, lane t receives column t):
// no rounding arithmetic, no per-element packing, half the gathered bytes, and the products run on the bf16 pipe --
// v_mfma_f32_16x16x32_bf16 takes 32 pairs per instruction where the fp32-row walk (spconv.hip spconv_wgrad_k) spends eight
// v_mfma_f32_16x16x4_f32 (16x the matrix-pipe time) or rounds and packs 8 x (NG + NX) values per lane and trip.
// Work split as in spconv_wgrad_k: a wave walks the pairs of one (offset, dy-row tile) range, four waves of a workgroup take four
// consecutive ranges and add their accumulators through LDS in a fixed order; partial blocks are summed by a second launch
// (deterministic, no atomics).  Channels stay in the shadows' fragment order inside the kernel (position p of a 32-channel group =
// channel 16 ((p >> 2) & 1) + 4 (p >> 3) + (p & 3)); the reduce launch undoes it when it writes dW[co][k][ci].
// Replaces spconv's implicit-GEMM weight gradient behind unidet3d/spconv_unet.py:43-56,148-154,178-183 for 32 / 64 channels.
#include "u3d_common.h"

namespace u3d {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8w;
typedef __attribute__((ext_vector_type(4))) short s16x4w;
typedef __attribute__((ext_vector_type(8))) short s16x8w;

struct WgRowsParams {
    const void* x;            // bf16 [n_x][Cs]
    const void* dy;           // bf16 [n_dy][Cd]
    const int32_t* rows_x;
    const int32_t* rows_dy;
    const int32_t* ts;        // tile_starts [K][n_tiles + 1] over the dy-side rows
    float* partial;           // [K][ceil(n_tiles / 4)][Cd * Cs] in (co position, ci position) order
    int K;
    int64_t cap;
    int n_tiles;
    int64_t n_x, n_dy;
};

// position of a channel inside the shadow's 32-channel groups and back
__host__ __device__ inline int frag_channel_of_pos(int p) { return (p & ~31) | (((p >> 2) & 1) << 4) | (((p >> 3) & 3) << 2) | (p & 3); }

template <int NG, int NX>      // Cd / 16, Cs / 16
__global__ __launch_bounds__(256) void spconv_wgrad_rows_k(WgRowsParams p) {
    constexpr int CD = NG * 16, CS = NX * 16;
    constexpr int LG = CD / 8, LX = CS / 8;              // lanes (16-byte pieces) per row
    constexpr int RG = 64 / LG, RX = 64 / LX;            // rows per load instruction
    constexpr int NIG = 32 / RG, NIX = 32 / RX;          // load instructions per 32-pair trip
    constexpr int TG = 32 * CD, TX = 32 * CS;            // bf16 elements of a tile
    constexpr int E = CD * CS;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, t16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __bf16* tg = reinterpret_cast<__bf16*>(smem_raw) + wave * (TG + TX);       // this wave's dy tile [32][CD], then its x tile [32][CS]
    __bf16* tx = tg + TG;

    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int k = j % p.K;
    const int range = ((j / p.K) * 8 + xcd) * 4 + wave;
    const bool active = range < p.n_tiles;               // wave-uniform; an idle wave still joins the reduction barriers
    const int lo = active ? p.ts[(int64_t)k * (p.n_tiles + 1) + range] : 0;
    const int hi = active ? p.ts[(int64_t)k * (p.n_tiles + 1) + range + 1] : 0;
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, p.n_x * CS * 2), rs_g = make_rsrc(p.dy, p.n_dy * CD * 2);
    const __amdgpu_buffer_rsrc_t rs_rx = make_rsrc(p.rows_x, (int64_t)p.K * p.cap * 4), rs_rg = make_rsrc(p.rows_dy, (int64_t)p.K * p.cap * 4);
    const int ksoff = (int)(k * p.cap) * 4;
    const int rg = lane / LG, pg = lane % LG, rx = lane / LX, px = lane % LX;

    f32x4 acc[NG][NX];
#pragma unroll
    for (int a = 0; a < NG; ++a)
#pragma unroll
        for (int b = 0; b < NX; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // indices of a trip (the lane's rows of every load instruction); pairs past the end are clamped into the range and their dy rows zeroed
    auto load_idx = [&](int (&ig)[NIG], int (&ix)[NIX], int base) {
#pragma unroll
        for (int i = 0; i < NIG; ++i) ig[i] = bload32(rs_rg, min(base + i * RG + rg, hi - 1) * 4, ksoff);
#pragma unroll
        for (int i = 0; i < NIX; ++i) ix[i] = bload32(rs_rx, min(base + i * RX + rx, hi - 1) * 4, ksoff);
    };
    auto load_rows = [&](f32x4 (&vg)[NIG], f32x4 (&vx)[NIX], const int (&ig)[NIG], const int (&ix)[NIX]) {
#pragma unroll
        for (int i = 0; i < NIG; ++i) vg[i] = bload128(rs_g, (int)__umul24(ig[i], CD * 2) + pg * 16, 0);
#pragma unroll
        for (int i = 0; i < NIX; ++i) vx[i] = bload128(rs_x, (int)__umul24(ix[i], CS * 2) + px * 16, 0);
    };
    auto compute = [&](f32x4 (&vg)[NIG], const f32x4 (&vx)[NIX], int base) {
        // rows -> private LDS tiles (row-major), dy rows of pairs past the end as zeros
#pragma unroll
        for (int i = 0; i < NIG; ++i) {
            if (base + i * RG + rg >= hi) vg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(tg + (i * RG + rg) * CD + pg * 8) = vg[i];
        }
#pragma unroll
        for (int i = 0; i < NIX; ++i) *reinterpret_cast<f32x4*>(tx + (i * RX + rx) * CS + px * 8) = vx[i];
        // fragments: lane (t16, q) <- column t16 of the [8 pairs 8q .. 8q+7][16 channels] block, two transpose reads of 4 pairs each
        bf16x8w fa[NG], fb[NX];
#pragma unroll
        for (int a = 0; a < NG; ++a) {
            const __bf16* s0 = tg + (8 * q + (t16 >> 2)) * CD + a * 16 + 4 * (t16 & 3);
            const s16x4w r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w*)s0);
            const s16x4w r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w*)(s0 + 4 * CD));
            fa[a] = __builtin_bit_cast(bf16x8w, s16x8w{r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]});
        }
#pragma unroll
        for (int b = 0; b < NX; ++b) {
            const __bf16* s0 = tx + (8 * q + (t16 >> 2)) * CS + b * 16 + 4 * (t16 & 3);
            const s16x4w r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w*)s0);
            const s16x4w r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w*)(s0 + 4 * CS));
            fb[b] = __builtin_bit_cast(bf16x8w, s16x8w{r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]});
        }
#pragma unroll
        for (int a = 0; a < NG; ++a)
#pragma unroll
            for (int b = 0; b < NX; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    };

    // walk: rows of trip t+1 and indices of trip t+2 are in flight while trip t is multiplied (two register sets, unrolled by two)
    const int ntrip = (hi - lo + 31) >> 5;
    if (ntrip > 0) {
        int igA[NIG], ixA[NIX], igB[NIG], ixB[NIX];
        f32x4 vgA[NIG], vxA[NIX], vgB[NIG], vxB[NIX];
        load_idx(igA, ixA, lo);
        load_idx(igB, ixB, lo + 32);
        load_rows(vgA, vxA, igA, ixA);
        for (int t = 0; t < ntrip; t += 2) {
            load_rows(vgB, vxB, igB, ixB);                       // rows of trip t+1 (clamped indices: always legal)
            load_idx(igA, ixA, lo + (t + 2) * 32);               // indices of trip t+2
            compute(vgA, vxA, lo + t * 32);
            if (t + 1 >= ntrip) break;
            load_rows(vgA, vxA, igA, ixA);                       // rows of trip t+2
            load_idx(igB, ixB, lo + (t + 3) * 32);
            compute(vgB, vxB, lo + (t + 1) * 32);
        }
    }

    // ((w0 + w2) + (w1 + w3)) through LDS, fixed order; accumulator element r of lane l in tile (a, b) is
    // dW_k[co position 16 a + 4 (l >> 4) + r][ci position 16 b + (l & 15)]
    __syncthreads();                                             // every wave is done with its tiles: the buffer becomes red[2][E]
    float* red = reinterpret_cast<float*>(smem_raw);
    float* mine = red + (wave & 1) * E;
    if (wave < 2) {
#pragma unroll
        for (int a = 0; a < NG; ++a)
#pragma unroll
            for (int b = 0; b < NX; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[(16 * a + 4 * q + r) * CS + 16 * b + t16] = acc[a][b][r];
    }
    __syncthreads();
    if (wave >= 2) {
#pragma unroll
        for (int a = 0; a < NG; ++a)
#pragma unroll
            for (int b = 0; b < NX; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[(16 * a + 4 * q + r) * CS + 16 * b + t16] += acc[a][b][r];
    }
    __syncthreads();
    if ((range & ~3) >= p.n_tiles) return;                       // (wave-uniform per workgroup: range & ~3 is its first range)
    float* out = p.partial + ((int64_t)k * ceil_div(p.n_tiles, 4) + (range >> 2)) * E;
    for (int e = tid; e < E; e += 256) out[e] = red[e] + red[E + e];
}

// dW[co][k][ci] = sum over the workgroup blocks of offset k (fixed order), positions mapped back to channels
__global__ __launch_bounds__(256) void wgrad_rows_reduce_k(const float* __restrict__ partial, int n_blocks, int K, int CD, int CS, float* __restrict__ dW) {
    const int E = CD * CS, k = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const float* src = partial + (int64_t)k * n_blocks * E + e;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    int r = 0;
    for (; r + 4 <= n_blocks; r += 4) {
        v0 += src[(int64_t)r * E];
        v1 += src[(int64_t)(r + 1) * E];
        v2 += src[(int64_t)(r + 2) * E];
        v3 += src[(int64_t)(r + 3) * E];
    }
    for (; r < n_blocks; ++r) v0 += src[(int64_t)r * E];
    const int co = frag_channel_of_pos(e / CS), ci = frag_channel_of_pos(e % CS);
    dW[((int64_t)co * K + k) * CS + ci] = (v0 + v1) + (v2 + v3);
}

template <int NG, int NX>
static int launch_wgrad_rows(const WgRowsParams& p, float* dW, hipStream_t s) {
    constexpr int CD = NG * 16, CS = NX * 16;
    constexpr int tiles = 4 * 32 * (CD + CS) * 2, red = 2 * CD * CS * 4;
    constexpr int lds = tiles > red ? tiles : red;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&spconv_wgrad_rows_k<NG, NX>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int64_t groups = ceil_div(ceil_div(p.n_tiles, 4), 8) * 8;
    hipLaunchKernelGGL((spconv_wgrad_rows_k<NG, NX>), dim3((unsigned)(groups * p.K)), dim3(256), lds, s, p);
    hipLaunchKernelGGL(wgrad_rows_reduce_k, dim3((unsigned)ceil_div(CD * CS, 256), p.K), dim3(256), 0, s, (const float*)p.partial,
                       (int)ceil_div(p.n_tiles, 4), p.K, CD, CS, dW);
    return check_launch("spconv_wgrad_rows");
}

}  // namespace u3d

using namespace u3d;

extern "C" {

int u3d_spconv_wgrad_rows_supported(int Cs, int Cd) { return (Cs == 32 || Cs == 64) && (Cd == 32 || Cd == 64) ? 1 : 0; }

int u3d_spconv_wgrad_rows(const void* x_bf16, int64_t n_rows_x, const void* dy_bf16, const int32_t* rows_x, const int32_t* rows_dy,
                          const int32_t* tile_starts, int K, int64_t cap, int64_t n_rows_dy, int tile_rows, int Cs, int Cd,
                          float* dW, void* ws, double flops_hint, u3d_stream_t stream) {
    if (!x_bf16 || !dy_bf16 || !rows_x || !rows_dy || !tile_starts || !dW || !ws || K <= 0 || cap <= 0 || n_rows_dy <= 0 || n_rows_x <= 0) return U3D_EINVAL;
    if (!u3d_spconv_wgrad_rows_supported(Cs, Cd)) { set_error("spconv_wgrad_rows: no instantiation for Cs=%d Cd=%d", Cs, Cd); return U3D_EUNSUPPORTED; }
    if (n_rows_x >= (1 << 24) || n_rows_dy >= (1 << 24) || (int64_t)K * cap * 4 >= 0x7fffffffLL) {
        set_error("spconv_wgrad_rows: %lld / %lld rows exceed the kernel's 32-bit addressing", (long long)n_rows_x, (long long)n_rows_dy);
        return U3D_EUNSUPPORTED;
    }
    if (tile_rows != u3d_spconv_wgrad_tile_rows(K, n_rows_dy, Cs, Cd)) { set_error("spconv_wgrad_rows: tile_rows %d does not match the plan", tile_rows); return U3D_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_CONV_WGRAD, s, flops_hint);
    WgRowsParams p;
    p.x = x_bf16; p.dy = dy_bf16; p.rows_x = rows_x; p.rows_dy = rows_dy; p.ts = tile_starts; p.partial = (float*)ws; p.K = K; p.cap = cap;
    p.n_tiles = (int)ceil_div(n_rows_dy, tile_rows); p.n_x = n_rows_x; p.n_dy = n_rows_dy;
    if (Cs == 32 && Cd == 32) return launch_wgrad_rows<2, 2>(p, dW, s);
    if (Cs == 64 && Cd == 32) return launch_wgrad_rows<2, 4>(p, dW, s);
    if (Cs == 32 && Cd == 64) return launch_wgrad_rows<4, 2>(p, dW, s);
    return launch_wgrad_rows<4, 4>(p, dW, s);
}

}  // extern "C"
