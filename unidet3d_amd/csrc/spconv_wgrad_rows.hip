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

// NP = 1: bf16 rows (the shadows, fragment order) -- what is instantiated.  NP = 3 (fp32 rows in natural order, every gathered value
// split exactly into three bf16 planes once on its way into LDS, six MFMAs per block pair, low-order products in accumulators of
// their own) was built and measured for the fp32 path in round 4 and is NOT instantiated: correct (the kernel tests passed at the
// fp32 bounds) but slower than the strided-fragment fp32 MFMA walk of spconv.hip -- 32 x 32 channels 152 us against 141, 64 x 64
// 176 against 123 (three plane tiles per operand: 49 / 98 KB of LDS per workgroup, 3x the LDS traffic, the split's VALU time).
// So was the walk on fp32 rows with native fp32 MFMAs (fp32 tiles, ds_read_b32 operands): 166 us against 139 (32 x 32), 193 against
// 125 (64 x 64) -- with 4-byte elements the tiles cost the occupancy the 32-cycle fp32 MFMAs need (48 / 80 KB per workgroup), and
// a K-step covers 4 pairs instead of 32.  The whole-row walk pays for 2-byte rows only.
template <int NG, int NX, int NP>      // Cd / 16, Cs / 16
__global__ __launch_bounds__(256) void spconv_wgrad_rows_k(WgRowsParams p) {
    constexpr int CD = NG * 16, CS = NX * 16;
    constexpr int EB = NP == 1 ? 2 : 4;                  // bytes per element of a gathered row
    constexpr int LG = CD * EB / 16, LX = CS * EB / 16;  // lanes (16-byte pieces) per row
    constexpr int RG = 64 / LG, RX = 64 / LX;            // rows per load instruction
    constexpr int NIG = 32 / RG, NIX = 32 / RX;          // load instructions per 32-pair trip
    constexpr int TG = 32 * CD, TX = 32 * CS;            // bf16 elements of one plane tile
    constexpr int E = CD * CS;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, t16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __bf16* tg = reinterpret_cast<__bf16*>(smem_raw) + wave * NP * (TG + TX);  // this wave's dy plane tiles [NP][32][CD], then its x plane tiles [NP][32][CS]
    __bf16* tx = tg + NP * TG;

    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int k = j % p.K;
    const int range = ((j / p.K) * 8 + xcd) * 4 + wave;
    const bool active = range < p.n_tiles;               // wave-uniform; an idle wave still joins the reduction barriers
    const int lo = active ? p.ts[(int64_t)k * (p.n_tiles + 1) + range] : 0;
    const int hi = active ? p.ts[(int64_t)k * (p.n_tiles + 1) + range + 1] : 0;
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, p.n_x * CS * EB), rs_g = make_rsrc(p.dy, p.n_dy * CD * EB);
    const __amdgpu_buffer_rsrc_t rs_rx = make_rsrc(p.rows_x, (int64_t)p.K * p.cap * 4), rs_rg = make_rsrc(p.rows_dy, (int64_t)p.K * p.cap * 4);
    const int ksoff = (int)(k * p.cap) * 4;
    const int rg = lane / LG, pg = lane % LG, rx = lane / LX, px = lane % LX;

    f32x4 acc[NG][NX], accl[NP == 3 ? NG : 1][NP == 3 ? NX : 1];
#pragma unroll
    for (int a = 0; a < NG; ++a)
#pragma unroll
        for (int b = 0; b < NX; ++b) {
            acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (NP == 3) accl[a][b] = acc[a][b];
        }

    // indices of a trip (the lane's rows of every load instruction); pairs past the end are clamped into the range and their dy rows zeroed
    auto load_idx = [&](int (&ig)[NIG], int (&ix)[NIX], int base) {
#pragma unroll
        for (int i = 0; i < NIG; ++i) ig[i] = bload32(rs_rg, min(base + i * RG + rg, hi - 1) * 4, ksoff);
#pragma unroll
        for (int i = 0; i < NIX; ++i) ix[i] = bload32(rs_rx, min(base + i * RX + rx, hi - 1) * 4, ksoff);
    };
    auto load_rows = [&](f32x4 (&vg)[NIG], f32x4 (&vx)[NIX], const int (&ig)[NIG], const int (&ix)[NIX]) {
#pragma unroll
        for (int i = 0; i < NIG; ++i) vg[i] = bload128(rs_g, (int)__umul24(ig[i], CD * EB) + pg * 16, 0);
#pragma unroll
        for (int i = 0; i < NIX; ++i) vx[i] = bload128(rs_x, (int)__umul24(ix[i], CS * EB) + px * 16, 0);
    };
    // rows -> this wave's LDS tiles (row-major [pair][channel] bf16 per plane), dy rows of pairs past the end as zeros
    auto put = [&](__bf16* tile, int plane_elems, int row, int C, int piece, const f32x4& v) {
        if constexpr (NP == 1) {
            *reinterpret_cast<f32x4*>(tile + row * C + piece * 8) = v;          // 8 bf16 of the row
        } else {
            unsigned w[2][3];
            split3_pair(v[0], v[1], w[0][0], w[0][1], w[0][2]);
            split3_pair(v[2], v[3], w[1][0], w[1][1], w[1][2]);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                *reinterpret_cast<uint2*>(tile + pl * plane_elems + row * C + piece * 4) = make_uint2(w[0][pl], w[1][pl]);      // 4 channels of one plane
        }
    };
    auto stage = [&](f32x4 (&vg)[NIG], const f32x4 (&vx)[NIX], int base) {
#pragma unroll
        for (int i = 0; i < NIG; ++i) {
            if (base + i * RG + rg >= hi) vg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            put(tg, TG, i * RG + rg, CD, pg, vg[i]);
        }
#pragma unroll
        for (int i = 0; i < NIX; ++i) put(tx, TX, i * RX + rx, CS, px, vx[i]);
    };
    // fragment of a [32 pairs][16 channels] block of a plane tile: lane (t16, q) <- column t16 over pairs 8q .. 8q+7, two transpose reads
    auto frag = [&](const __bf16* tile, int C, int blk) -> bf16x8w {
        const __bf16* s0 = tile + (8 * q + (t16 >> 2)) * C + blk * 16 + 4 * (t16 & 3);
        const s16x4w r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w*)s0);
        const s16x4w r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w*)(s0 + 4 * C));
        return __builtin_bit_cast(bf16x8w, s16x8w{r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]});
    };
    auto multiply = [&]() {
        bf16x8w fb[NX][NP];
#pragma unroll
        for (int b = 0; b < NX; ++b)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) fb[b][pl] = frag(tx + pl * TX, CS, b);
#pragma unroll
        for (int a = 0; a < NG; ++a) {
            bf16x8w fa[NP];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) fa[pl] = frag(tg + pl * TG, CD, a);
#pragma unroll
            for (int b = 0; b < NX; ++b) {
                if constexpr (NP == 3) {
#pragma unroll
                    for (int o = 2; o >= 1; --o)
#pragma unroll
                        for (int qa = 0; qa <= o; ++qa)
                            accl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[qa], fb[b][o - qa], accl[a][b], 0, 0, 0);
                }
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[b][0], acc[a][b], 0, 0, 0);
            }
        }
    };

    // walk: ONE register set of rows -- a trip's rows go to LDS first, then the next trip's loads are issued into the same registers
    // and land while the fragments are read and multiplied; indices two trips ahead (two index sets, unrolled by two)
    const int ntrip = (hi - lo + 31) >> 5;
    if (ntrip > 0) {
        int igA[NIG], ixA[NIX], igB[NIG], ixB[NIX];
        f32x4 vg[NIG], vx[NIX];
        load_idx(igA, ixA, lo);
        load_idx(igB, ixB, lo + 32);
        load_rows(vg, vx, igA, ixA);
        for (int t = 0; t < ntrip; t += 2) {
            stage(vg, vx, lo + t * 32);
            load_rows(vg, vx, igB, ixB);                         // rows of trip t+1 (clamped indices: always legal)
            load_idx(igA, ixA, lo + (t + 2) * 32);               // indices of trip t+2
            multiply();
            if (t + 1 >= ntrip) break;
            stage(vg, vx, lo + (t + 1) * 32);
            load_rows(vg, vx, igA, ixA);                         // rows of trip t+2
            load_idx(igB, ixB, lo + (t + 3) * 32);
            multiply();
        }
    }
    if constexpr (NP == 3) {
#pragma unroll
        for (int a = 0; a < NG; ++a)
#pragma unroll
            for (int b = 0; b < NX; ++b) acc[a][b] += accl[a][b];
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

// Wider layers (96 .. 160 channels): the accumulator block no longer fits one wave, so the four waves of a workgroup share ONE walk:
// the workgroup gathers the 32 rows of a trip together (16-byte pieces dealt over the 256 threads) into tiles all four waves read,
// wave (wa, wb) keeps the (co half wa) x (ci half wb) quarter of dW_k in its accumulators, one barrier per trip (two tile buffers).
// The workgroup walks the four ranges spconv_wgrad_rows_k would give its four waves as one longer range (they belong to the same
// offset), so the partial-block layout and the reduce launch are the same and no LDS reduction is needed.
template <int NG, int NX>      // Cd / 16, Cs / 16 (both even)
__global__ __launch_bounds__(256) void spconv_wgrad_rows_coop_k(WgRowsParams p) {
    constexpr int CD = NG * 16, CS = NX * 16, NGW = NG / 2, NXW = NX / 2;
    constexpr int LG = CD / 8, LX = CS / 8;              // 16-byte pieces per row
    constexpr int PG = 32 * LG, PX = 32 * LX;            // pieces per trip
    constexpr int NPG = (PG + 255) / 256, NPX = (PX + 255) / 256;
    constexpr int TG = 32 * CD, TX = 32 * CS;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __bf16* tiles = reinterpret_cast<__bf16*>(smem_raw);         // [2 buffers][dy tile [32][CD] | x tile [32][CS]]
    const int tid = threadIdx.x, lane = tid & 63, t16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave & 1, wb = wave >> 1;

    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int k = j % p.K;
    const int group = (j / p.K) * 8 + xcd;                       // four consecutive dy-row tiles
    const int n_groups = (int)ceil_div(p.n_tiles, 4);
    if (group >= n_groups) return;                               // workgroup-uniform
    const int lo = p.ts[(int64_t)k * (p.n_tiles + 1) + group * 4];
    const int hi = p.ts[(int64_t)k * (p.n_tiles + 1) + min(group * 4 + 4, p.n_tiles)];
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, p.n_x * CS * 2), rs_g = make_rsrc(p.dy, p.n_dy * CD * 2);
    const __amdgpu_buffer_rsrc_t rs_rx = make_rsrc(p.rows_x, (int64_t)p.K * p.cap * 4), rs_rg = make_rsrc(p.rows_dy, (int64_t)p.K * p.cap * 4);
    const int ksoff = (int)(k * p.cap) * 4;

    f32x4 acc[NGW][NXW];
#pragma unroll
    for (int a = 0; a < NGW; ++a)
#pragma unroll
        for (int b = 0; b < NXW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_idx = [&](int (&ig)[NPG], int (&ix)[NPX], int base) {
#pragma unroll
        for (int i = 0; i < NPG; ++i) ig[i] = bload32(rs_rg, min(base + (tid + i * 256) / LG, hi - 1) * 4, ksoff);
#pragma unroll
        for (int i = 0; i < NPX; ++i) ix[i] = bload32(rs_rx, min(base + (tid + i * 256) / LX, hi - 1) * 4, ksoff);
    };
    auto load_rows = [&](f32x4 (&vg)[NPG], f32x4 (&vx)[NPX], const int (&ig)[NPG], const int (&ix)[NPX]) {
#pragma unroll
        for (int i = 0; i < NPG; ++i) vg[i] = bload128(rs_g, (int)__umul24(ig[i], CD * 2) + ((tid + i * 256) % LG) * 16, 0);
#pragma unroll
        for (int i = 0; i < NPX; ++i) vx[i] = bload128(rs_x, (int)__umul24(ix[i], CS * 2) + ((tid + i * 256) % LX) * 16, 0);
    };
    auto stage = [&](f32x4 (&vg)[NPG], const f32x4 (&vx)[NPX], int base, int buf) {
        __bf16* tg = tiles + buf * (TG + TX);
        __bf16* tx = tg + TG;
#pragma unroll
        for (int i = 0; i < NPG; ++i) {
            const int pid = tid + i * 256;
            if (pid < PG) {
                if (base + pid / LG >= hi) vg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(tg + pid * 8) = vg[i];          // row-major [32][CD]: piece pid = (row, piece) in order
            }
        }
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            const int pid = tid + i * 256;
            if (pid < PX) *reinterpret_cast<f32x4*>(tx + pid * 8) = vx[i];
        }
    };
    auto frag = [&](const __bf16* tile, int C, int blk) -> bf16x8w {
        const __bf16* s0 = tile + (8 * q + (t16 >> 2)) * C + blk * 16 + 4 * (t16 & 3);
        const s16x4w r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w*)s0);
        const s16x4w r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4w*)(s0 + 4 * C));
        return __builtin_bit_cast(bf16x8w, s16x8w{r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]});
    };
    auto multiply = [&](int buf) {
        const __bf16* tg = tiles + buf * (TG + TX);
        const __bf16* tx = tg + TG;
        bf16x8w fb[NXW];
#pragma unroll
        for (int b = 0; b < NXW; ++b) fb[b] = frag(tx, CS, wb * NXW + b);
#pragma unroll
        for (int a = 0; a < NGW; ++a) {
            const bf16x8w fa = frag(tg, CD, wa * NGW + a);
#pragma unroll
            for (int b = 0; b < NXW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[b], acc[a][b], 0, 0, 0);
        }
    };

    const int ntrip = (hi - lo + 31) >> 5;                        // workgroup-uniform: every wave runs the same barriers
    if (ntrip > 0) {
        int igA[NPG], ixA[NPX], igB[NPG], ixB[NPX];
        f32x4 vg[NPG], vx[NPX];
        load_idx(igA, ixA, lo);
        load_idx(igB, ixB, lo + 32);
        load_rows(vg, vx, igA, ixA);
        for (int t = 0; t < ntrip; t += 2) {
            stage(vg, vx, lo + t * 32, 0);
            load_rows(vg, vx, igB, ixB);
            load_idx(igA, ixA, lo + (t + 2) * 32);
            __syncthreads();                                     // buffer 0 complete; everyone is past multiply(buffer 0) of trip t-2
            multiply(0);
            if (t + 1 >= ntrip) break;
            stage(vg, vx, lo + (t + 1) * 32, 1);
            load_rows(vg, vx, igA, ixA);
            load_idx(igB, ixB, lo + (t + 3) * 32);
            __syncthreads();
            multiply(1);
        }
    }
    float* out = p.partial + ((int64_t)k * n_groups + group) * (CD * CS);
#pragma unroll
    for (int a = 0; a < NGW; ++a)
#pragma unroll
        for (int b = 0; b < NXW; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(16 * (wa * NGW + a) + 4 * q + r) * CS + 16 * (wb * NXW + b) + t16] = acc[a][b][r];
}

// dW[co][k][ci] = sum over the workgroup blocks of offset k (fixed order), positions mapped back to channels
__global__ __launch_bounds__(256) void wgrad_rows_reduce_k(const float* __restrict__ partial, int n_blocks, int K, int CD, int CS, int frag_order,
                                                           float* __restrict__ dW) {
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
    const int co = frag_order ? frag_channel_of_pos(e / CS) : e / CS, ci = frag_order ? frag_channel_of_pos(e % CS) : e % CS;
    dW[((int64_t)co * K + k) * CS + ci] = (v0 + v1) + (v2 + v3);
}

template <int NG, int NX, int NP>
static int launch_wgrad_rows(const WgRowsParams& p, float* dW, hipStream_t s) {
    constexpr int CD = NG * 16, CS = NX * 16;
    constexpr int tiles = 4 * NP * 32 * (CD + CS) * 2, red = 2 * CD * CS * 4;
    constexpr int lds = tiles > red ? tiles : red;
    static DeviceOnce attr_set;              // the attribute is per DEVICE
    int dev = 0;
    hipGetDevice(&dev);
    if (attr_set.needed(dev)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&spconv_wgrad_rows_k<NG, NX, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set.done(dev);
    }
    const int64_t groups = ceil_div(ceil_div(p.n_tiles, 4), 8) * 8;
    hipLaunchKernelGGL((spconv_wgrad_rows_k<NG, NX, NP>), dim3((unsigned)(groups * p.K)), dim3(256), lds, s, p);
    hipLaunchKernelGGL(wgrad_rows_reduce_k, dim3((unsigned)ceil_div(CD * CS, 256), p.K), dim3(256), 0, s, (const float*)p.partial,
                       (int)ceil_div(p.n_tiles, 4), p.K, CD, CS, NP == 1 ? 1 : 0, dW);
    return check_launch("spconv_wgrad_rows");
}

template <int NG, int NX>
static int launch_wgrad_rows_coop(const WgRowsParams& p, float* dW, hipStream_t s) {
    constexpr int CD = NG * 16, CS = NX * 16;
    constexpr int lds = 2 * 32 * (CD + CS) * 2;
    static DeviceOnce attr_set;              // the attribute is per DEVICE
    int dev = 0;
    hipGetDevice(&dev);
    if (attr_set.needed(dev)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&spconv_wgrad_rows_coop_k<NG, NX>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set.done(dev);
    }
    const int64_t groups = ceil_div(ceil_div(p.n_tiles, 4), 8) * 8;
    hipLaunchKernelGGL((spconv_wgrad_rows_coop_k<NG, NX>), dim3((unsigned)(groups * p.K)), dim3(256), lds, s, p);
    hipLaunchKernelGGL(wgrad_rows_reduce_k, dim3((unsigned)ceil_div(CD * CS, 256), p.K), dim3(256), 0, s, (const float*)p.partial,
                       (int)ceil_div(p.n_tiles, 4), p.K, CD, CS, 1, dW);
    return check_launch("spconv_wgrad_rows_coop");
}

}  // namespace u3d

using namespace u3d;

extern "C" {

static bool rows_wave(int Cs, int Cd) { return (Cs == 32 || Cs == 64) && (Cd == 32 || Cd == 64); }
static bool rows_coop(int Cs, int Cd) {
    static const int combos[][2] = {{96, 96}, {128, 128}, {160, 160}, {64, 96}, {96, 64}, {96, 128}, {128, 96}, {128, 160}, {160, 128}, {128, 64}, {192, 96}, {256, 128}};
    for (auto& c : combos) if (c[0] == Cs && c[1] == Cd) return true;
    return false;
}
int u3d_spconv_wgrad_rows_supported(int Cs, int Cd) { return rows_wave(Cs, Cd) || rows_coop(Cs, Cd) ? 1 : 0; }

int u3d_spconv_wgrad_rows(const void* x_bf16, int64_t n_rows_x, const void* dy_bf16, const int32_t* rows_x, const int32_t* rows_dy,
                          const int32_t* tile_starts, int K, int64_t cap, int64_t n_rows_dy, int tile_rows, int Cs, int Cd,
                          float* dW, void* ws, double flops_hint, u3d_stream_t stream) {
    if (!x_bf16 || !dy_bf16 || !rows_x || !rows_dy || !tile_starts || !dW || !ws || K <= 0 || cap <= 0 || n_rows_dy <= 0 || n_rows_x <= 0) return U3D_EINVAL;
    if (!u3d_spconv_wgrad_rows_supported(Cs, Cd)) { set_error("spconv_wgrad_rows: no instantiation for Cs=%d Cd=%d", Cs, Cd); return U3D_EUNSUPPORTED; }
    if (n_rows_x >= (1 << 24) || n_rows_dy >= (1 << 24) || (int64_t)K * cap * 4 >= 0x7fffffffLL ||
        n_rows_x * Cs * 2 >= 0x7fffffffLL || n_rows_dy * Cd * 2 >= 0x7fffffffLL) {        // rows are addressed as (int)__umul24(row, C * 2)
        set_error("spconv_wgrad_rows: %lld / %lld rows exceed the kernel's 32-bit addressing", (long long)n_rows_x, (long long)n_rows_dy);
        return U3D_EUNSUPPORTED;
    }
    if (tile_rows != u3d_spconv_wgrad_tile_rows(K, n_rows_dy, Cs, Cd)) { set_error("spconv_wgrad_rows: tile_rows %d does not match the plan", tile_rows); return U3D_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_CONV_WGRAD, s, flops_hint);
    WgRowsParams p;
    p.x = x_bf16; p.dy = dy_bf16; p.rows_x = rows_x; p.rows_dy = rows_dy; p.ts = tile_starts; p.partial = (float*)ws; p.K = K; p.cap = cap;
    p.n_tiles = (int)ceil_div(n_rows_dy, tile_rows); p.n_x = n_rows_x; p.n_dy = n_rows_dy;
    if (Cs == 32 && Cd == 32) return launch_wgrad_rows<2, 2, 1>(p, dW, s);
    if (Cs == 64 && Cd == 32) return launch_wgrad_rows<2, 4, 1>(p, dW, s);
    if (Cs == 32 && Cd == 64) return launch_wgrad_rows<4, 2, 1>(p, dW, s);
    if (Cs == 64 && Cd == 64) return launch_wgrad_rows<4, 4, 1>(p, dW, s);
#define U3D_COOP(cs, cd) if (Cs == cs && Cd == cd) return launch_wgrad_rows_coop<cd / 16, cs / 16>(p, dW, s);
    U3D_COOP(96, 96) U3D_COOP(128, 128) U3D_COOP(160, 160)                  // SubM blocks of levels 3-5
    U3D_COOP(64, 96) U3D_COOP(96, 64) U3D_COOP(96, 128) U3D_COOP(128, 96) U3D_COOP(128, 160) U3D_COOP(160, 128)      // strided / inverse
    U3D_COOP(128, 64) U3D_COOP(192, 96) U3D_COOP(256, 128)                    // first convolution of the tail blocks (skip concatenation)
#undef U3D_COOP
    set_error("spconv_wgrad_rows: no instantiation for Cs=%d Cd=%d", Cs, Cd);
    return U3D_EUNSUPPORTED;
}

}  // extern "C"
