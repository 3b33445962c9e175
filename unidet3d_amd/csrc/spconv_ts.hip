// Tile-stationary submanifold convolution (round 6): forward and input gradient of SubMConv3d(k = 3) with the accumulators of a
// row tile in REGISTERS for all 27 offsets and all output columns, and every source row a tile needs fetched ONCE.
//
// The pair-list kernels (spconv.hip wave tiles, spconv_wg.hip workgroup tiles) walk an offset's (in, out) pairs: every pair gathers
// its source row again (11.9 gathers per row at cfg2 level 1, once more per 32-column slice of C_out), splits it into the three bf16
// planes again, turns it through an LDS image, and adds the product into an LDS accumulator tile by read-modify-write.  Measured
// (DESIGN.md 4.12): the texture path (14.5 cycles per gathered line that misses L1), the split's VALU time, the LDS staging and the
// accumulator traffic each cost 10-20 % and add up.
//
// Here a workgroup (four waves) owns T = 64 RT consecutive rows of the canonical order.  The rulebook side (subm_halo_k below, once
// per level and step, reused by the ~17 launches of that level) gives per tile
//   * halo[tile][..]  : the sorted UNIQUE source rows of the tile's 27 T (row, offset) neighbours -- 2.9-3.8 T rows, not 11.9 T;
//   * loc[tile][k][r] : the position of neighbour k of row r in that list (uint16, 0xFFFF = no neighbour);
//   * pmask[tile][p]  : the offsets that have a neighbour among the p-th H entries of the list.
// The kernel streams the halo through LDS in passes of H rows: each row is loaded once (whole 128-byte lines), split once into
// its three bf16 planes (u3d_common.h "bf16x3") and stored in MFMA fragment order; then, offset by offset, every wave forms the B
// operand of its 16-row sub-tiles by one indexed ds_read_b128 per plane (rows without a neighbour in this pass read a zero row),
// the A operand -- the offset's packed weights, staged once per workgroup in LDS as in spconv_wg.hip -- by ds_read_b128, and issues
// six v_mfma_f32_16x16x32_bf16 per (16 rows x 16 columns x 32 channels) straight into the tile's accumulators (h.h into the running
// tile, the five low-order plane products into a second one: the bf16 MFMA truncates products against C).  A 16-row
// sub-tile without any neighbour for the offset (wave ballot) is skipped.  No pair lists, no scatter index, no accumulator tile in
// LDS, no per-pair split, no offset-group partial buffers and no reduce launch.  The price is matrix work on the zero rows: 1.7-2x
// the pair count at cfg2 (tools/halo_stats.py) on a matrix pipe the pair kernels leave ~85 % idle.
//
// The input gradient is the same launch with the offsets mirrored (SubM pairs are symmetric: in = out + d_k  <=>  the source of
// gradient row i at offset k is its neighbour 26 - k) and the transposed weight pack.
// Replaces spconv's implicit-GEMM SubMConv3d behind unidet3d/spconv_unet.py:43-56 (reference); results differ from the pair kernels
// only in fp32 summation order (offsets ascending per pass instead of ascending overall).
#include <limits.h>
#include <stdlib.h>

#include "u3d_common.h"

// -DU3D_TS_ABL=<mask>: timing ablations (tools/build_variant.sh), WRONG results by construction.  1: no MFMAs, 2: no halo row loads /
// split / LDS writes, 4: no weight staging, 8: no barriers inside the offset loop, 16: no split arithmetic, 32: no fragment reads
#ifndef U3D_TS_ABL
#define U3D_TS_ABL 0
#endif

namespace u3d {

typedef __attribute__((ext_vector_type(8))) __bf16 ts_bf16x8;

__device__ __forceinline__ f32x4 ts_mfma(const f32x4& a, const ts_bf16x8& b, const f32x4& c) {
#if U3D_TS_ABL & 1
    asm volatile("" :: "v"(a), "v"(b));
    return c;
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ts_bf16x8, a), b, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ void ts_barrier() {      // this wave's LDS traffic done, then the workgroup barrier; vmcnt is NOT drained
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---- rulebook side: per-tile unique source rows + local positions -------------------------------------------------------------
__device__ __forceinline__ int ts_lookup(const int32_t* __restrict__ coords, int64_t o, int k, const Index& ix) {
    const int dx = k / 9 - 1, dy = (k / 3) % 3 - 1, dz = k % 3 - 1;
    const int4 c = *reinterpret_cast<const int4*>(coords + o * 4);
    return index_lookup(ix, c.x, c.y + dx, c.z + dy, c.w + dz);
}

// One workgroup per tile of T rows.  NS: power of two >= 27 T (bitonic sort of the candidates in LDS).
template <int T, int NS>
__global__ __launch_bounds__(256) void subm_halo_k(const int32_t* __restrict__ coords, int64_t n, Index ix, int H, int pmax,
                                                   int32_t* __restrict__ nhalo, int32_t* __restrict__ halo, uint16_t* __restrict__ loc,
                                                   uint32_t* __restrict__ pmask) {
    constexpr int NC = 27 * T, CPT = (NC + 255) / 256, EP = NS / 256;
    static_assert(NS >= NC && EP <= 32, "sort size");
    __shared__ int srt[NS];
    __shared__ int ubuf[NC];
    __shared__ int wsum[4];
    __shared__ unsigned pm[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tile = blockIdx.x, r0 = tile * T;
    int cand[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int idx = tid + 256 * j;
        int v = -1;
        if (idx < NC) {
            const int k = idx / T, r = idx - k * T;
            const int64_t row = r0 + r;
            if (row < n) v = ts_lookup(coords, row, k, ix);
            srt[idx] = v >= 0 ? v : INT_MAX;
        }
        cand[j] = v;
    }
    for (int idx = NC + tid; idx < NS; idx += 256) srt[idx] = INT_MAX;
    if (tid < 64) pm[tid] = 0u;
    __syncthreads();
    for (int k = 2; k <= NS; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int pi = tid; pi < NS / 2; pi += 256) {
                const int i = ((pi & ~(j - 1)) << 1) | (pi & (j - 1));
                const int l = i | j;
                const int a = srt[i], b = srt[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { srt[i] = b; srt[l] = a; }
            }
            __syncthreads();
        }
    // unique: every thread flags its EP consecutive entries, block scan of the counts, compaction into ubuf
    const int i0 = tid * EP;
    int prev = i0 ? srt[i0 - 1] : -1;
    unsigned flags = 0u;
    int cnt = 0;
#pragma unroll
    for (int e = 0; e < EP; ++e) {
        const int v = srt[i0 + e];
        const bool f = v != INT_MAX && v != prev;
        flags |= (f ? 1u : 0u) << e;
        cnt += f ? 1 : 0;
        prev = v;
    }
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0, nh = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int t = wsum[w]; if (w < wave) woff += t; nh += t; }
    int pos = woff + incl - cnt;
#pragma unroll
    for (int e = 0; e < EP; ++e)
        if ((flags >> e) & 1u) ubuf[pos++] = srt[i0 + e];
    __syncthreads();
    for (int i = tid; i < nh; i += 256) halo[tile * NC + i] = ubuf[i];
    if (tid == 0) nhalo[tile] = nh;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int idx = tid + 256 * j;
        if (idx < NC) {
            const int k = idx / T, r = idx - k * T;
            const int v = cand[j];
            unsigned short o = 0xFFFFu;
            if (v >= 0) {
                int lo = 0, hi = nh;            // first entry >= v (v is in the list)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (ubuf[mid] < v) lo = mid + 1; else hi = mid;
                }
                o = (unsigned short)lo;
                atomicOr(&pm[lo / H], 1u << k);
            }
            loc[(tile * 27 + k) * T + r] = o;
        }
    }
    __syncthreads();
    if (tid < pmax) pmask[tile * pmax + tid] = pm[tid];
}

// ---- the convolution ---------------------------------------------------------------------------------------------------------------
struct TsParams {
    const float* src;
    const void* w;             // u3d_weight_pack_x3 layout: [slice][k][unit][column block][plane][lane] 16-byte vectors
    const int32_t* nhalo;
    const int32_t* halo;
    const uint16_t* loc;
    const uint32_t* pmask;
    const float* addend;
    float* out;
    int64_t n;
    int Cs, Cd;
    int n_tiles, pmax, flip;
    unsigned long long* trace;   // -DU3D_TS_TRACE builds: [n_tiles][32] cycle stamps of wave 0 (tools/ts_trace.py), else unused
};

constexpr int ts_halo_bytes(int cs32, int h) { return cs32 * 3 * (h + 1) * 64; }
constexpr int ts_wslot_bytes(int cs32, int cd16, int kg) { return kg * (cd16 / 2) * cs32 * 6144; }
constexpr int ts_loc_bytes(int rt) { return 27 * 64 * rt * 2; }      // the tile's loc table (uint16 [27][T])
constexpr int ts_lds_bytes(int cs32, int cd16, int rt, int h, int kg, int nslot) {
    return ts_halo_bytes(cs32, h) + ts_loc_bytes(rt) + nslot * ts_wslot_bytes(cs32, cd16, kg);
}
constexpr int ts_per_cu(int bytes) { return 160 * 1024 / bytes > 4 ? 4 : 160 * 1024 / bytes; }      // (registers allow four workgroups per CU at most)
// two weight slots (one barrier per step) unless the second slot costs a workgroup per CU (then one slot, two barriers)
constexpr int ts_nslot(int cs32, int cd16, int rt, int h, int kg) {
    return ts_per_cu(ts_lds_bytes(cs32, cd16, rt, h, kg, 2)) >= ts_per_cu(ts_lds_bytes(cs32, cd16, rt, h, kg, 1)) ? 2 : 1;
}

// byte offset of 16-byte chunk q of slot s inside a (unit, plane) image: 64-byte rows, chunk stored at q ^ ((-(s >> 2)) & 3) -- with
// ds_read_b128's lane groups ({0-3, 12-15, 20-27}, ...) sixteen consecutive slots are then conflict-free (MI355X_MICROARCH.md LDS)
__device__ __forceinline__ int ts_slot_off(int s, int q) { return (s << 6) | ((q ^ ((0 - (s >> 2)) & 3)) << 4); }

// CS32: 32-channel units of the source rows; CD16: 16-column blocks of the output; RT: 16-row sub-tiles per wave (T = 64 RT rows per
// workgroup); H: halo rows per pass; KG: consecutive offsets per step (1, or 3 = the dz triple of one (dx, dy): a third of the
// barriers and weight hand-overs); NSLOT: LDS copies of a step's weights (2: one barrier per step, 1: two)
template <int CS32, int CD16, int RT, int H, int KG, int NSLOT>
__global__ __launch_bounds__(256) void spconv_ts_k(TsParams p) {
    extern __shared__ __attribute__((aligned(16))) char ts_smem[];
    constexpr int T = 64 * RT, NC = 27 * T, NSL = CD16 / 2, NG = 27 / KG;
    constexpr int PLANE = (H + 1) * 64, HALO = ts_halo_bytes(CS32, H), WSLOT = ts_wslot_bytes(CS32, CD16, KG);
    constexpr int SLICE_P = KG * CS32 * 384;                       // 16-byte pieces of one slice's share of a step (contiguous in the pack)
    constexpr int NPIECE = WSLOT / 16, NP = (NPIECE + 255) / 256;
    constexpr int ROUNDS = H / 32;
    static_assert(H % 32 == 0 && NPIECE % 64 == 0 && 27 % KG == 0, "tile shapes");
    constexpr int LOCB = ts_loc_bytes(RT);
    char* const hl = ts_smem;
    char* const ll = ts_smem + HALO;              // loc table of the tile: global latency must not sit between two offsets
    char* const wl = ts_smem + HALO + LOCB;
    const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = (int)xcd_swizzle(blockIdx.x, gridDim.x);      // neighbouring tiles (shared halo rows) on one XCD / L2
    const int64_t r0 = (int64_t)tile * T;

    const __amdgpu_buffer_rsrc_t rs_src = make_rsrc(p.src, p.n * p.Cs * 4);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w);
    const __amdgpu_buffer_rsrc_t rs_h = make_rsrc(p.halo + (int64_t)tile * NC, (int64_t)NC * 4);
    const __amdgpu_buffer_rsrc_t rs_l = make_rsrc(p.loc + (int64_t)tile * NC, (int64_t)NC * 2);
    const int cs4 = p.Cs * 4;
    const int nh = __builtin_amdgcn_readfirstlane(p.nhalo[tile]);
    const int npass = (nh + H - 1) / H;
    const uint32_t* pmk = p.pmask + (int64_t)tile * p.pmax;

#ifdef U3D_TS_TRACE
    unsigned long long* const tr = p.trace ? p.trace + (int64_t)blockIdx.x * 32 : nullptr;
#define TS_STAMP(i) do { if (tr && tid == 0 && (i) < 32) tr[(i)] = clock64(); } while (0)
#else
#define TS_STAMP(i) do {} while (0)
#endif
    TS_STAMP(0);
    f32x4 hi[RT][CD16], lo[RT][CD16];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nb = 0; nb < CD16; ++nb) { hi[rt][nb] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[rt][nb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (tid < CS32 * 3 * 4) *reinterpret_cast<f32x4*>(hl + (tid >> 2) * PLANE + H * 64 + (tid & 3) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};      // the zero row

    // sub-tile rt of wave w = rows 64 rt + 16 w ..: neighbouring sub-tiles (similar neighbour patterns) go to DIFFERENT waves, so the
    // waves of a workgroup have about the same work between two barriers
    const int loc_lane = (wave * 16 + i16) * 2;                   // byte offset of this lane's entry for (table offset 0, sub-tile 0)
    {
        constexpr int LP = LOCB / 16;                             // 16-byte pieces of the table
#pragma unroll
        for (int i = 0; i < (LP + 255) / 256; ++i)
            if ((i + 1) * 256 <= LP || tid + i * 256 < LP) *reinterpret_cast<f32x4*>(ll + (tid + i * 256) * 16) = bload128(rs_l, (tid + i * 256) * 16, 0);
    }
    const int pc = tid & 7, prow = tid >> 3;                      // load phase: 16-byte piece of the row, row within a round of 32
    const int w_wr = tid * 16, w_rd = lane * 16;

    // weights travel global -> registers (one step ahead) -> LDS slot.  Two register sets filled two steps ahead measured SLOWER
    // (level 1, 32 -> 32: 161 -> 226 us): the steps are not waiting for the weights' latency
    f32x4 wreg[NP];
    auto load_w = [&](int g) {         // weights of step g (table offsets KG g .. KG g + KG - 1) -> registers; contiguous per slice in the pack
        if (U3D_TS_ABL & 4) return;
        const int kbase = p.flip ? 27 - KG * (g + 1) : KG * g;    // mirrored offsets: the same block, walked backwards
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int piece = tid + i * 256;                      // 16-byte piece of the slot
            if ((i + 1) * 256 <= NPIECE || piece < NPIECE) {
                const int sl = piece / SLICE_P, rem = piece - sl * SLICE_P;
                wreg[i] = bload128(rs_w, rem * 16 + ((sl * 27 + kbase) * CS32) * 6144, 0);
            }
        }
    };
    auto write_w = [&](int slot) {
        if (U3D_TS_ABL & 4) return;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if ((i + 1) * 256 <= NPIECE || tid + i * 256 < NPIECE) *reinterpret_cast<f32x4*>(wl + slot * WSLOT + w_wr + i * 4096) = wreg[i];
    };
    auto group_mask = [](unsigned km) -> unsigned {               // bit g: some offset of step g has work in this pass
        if constexpr (KG == 1) return km;
        unsigned gm = 0u;
#pragma unroll
        for (int g = 0; g < NG; ++g) gm |= ((km >> (KG * g)) & ((1u << KG) - 1u)) ? 1u << g : 0u;
        return gm;
    };
    auto next_bit = [](unsigned mask, int after) -> int {         // lowest set bit above `after` (32: none)
        const unsigned rest = after >= 31 ? 0u : mask & (~1u << after);
        return rest ? __builtin_ctz(rest) : 32;
    };

    int hrow[ROUNDS];                                             // source rows of the NEXT pass's halo slots (one pass ahead)
#pragma unroll
    for (int j = 0; j < ROUNDS; ++j) hrow[j] = bload32(rs_h, (j * 32 + prow) * 4, 0);
    unsigned km_next = npass ? pmk[0] : 0u;
    for (int ps = 0; ps < npass; ++ps) {
        const unsigned km = __builtin_amdgcn_readfirstlane(km_next);
        const unsigned gm = group_mask(km);
        if (ps + 1 < npass) km_next = pmk[ps + 1];
        const int cnt = min(H, nh - ps * H);
        TS_STAMP(1 + ps * 5);
        if (ps) ts_barrier();                                     // every wave is done with the previous pass's rows and weight slots
        TS_STAMP(2 + ps * 5);
        int g = gm ? __builtin_ctz(gm) : 32;
        if (g < 32) load_w(g);
        // ---- halo rows of this pass: global -> three bf16 planes in LDS (fragment order) ----
#if !(U3D_TS_ABL & 2)
#pragma unroll
        for (int j0 = 0; j0 < ROUNDS; j0 += 4) {
            f32x4 v[4][CS32];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j0 + j < ROUNDS) {
                    const bool ok = (j0 + j) * 32 + prow < cnt;
                    const int ro = ok ? (int)__umul24(hrow[j0 + j], cs4) + pc * 16 : 0x7ffffff0;      // past the end: the load returns zeros
#pragma unroll
                    for (int u = 0; u < CS32; ++u) v[j][u] = bload128(rs_src, ro, u * 128);
                    if (ps + 1 < npass) hrow[j0 + j] = bload32(rs_h, ((ps + 1) * H + (j0 + j) * 32 + prow) * 4, 0);
                }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j0 + j < ROUNDS) {
                    const int s = (j0 + j) * 32 + prow;
                    const int off = ts_slot_off(s, pc & 3) + (pc >> 2) * 8;
#pragma unroll
                    for (int u = 0; u < CS32; ++u) {
                        unsigned h0, m0, l0, h1, m1, l1;
#if U3D_TS_ABL & 16
                        h0 = m0 = __builtin_bit_cast(unsigned, v[j][u][0]); l0 = h1 = __builtin_bit_cast(unsigned, v[j][u][1]);
                        m1 = __builtin_bit_cast(unsigned, v[j][u][2]); l1 = __builtin_bit_cast(unsigned, v[j][u][3]);
#else
                        split3_pair(v[j][u][0], v[j][u][1], h0, m0, l0);
                        split3_pair(v[j][u][2], v[j][u][3], h1, m1, l1);
#endif
                        char* const b = hl + u * 3 * PLANE + off;
                        *reinterpret_cast<uint2*>(b) = make_uint2(h0, h1);
                        *reinterpret_cast<uint2*>(b + PLANE) = make_uint2(m0, m1);
                        *reinterpret_cast<uint2*>(b + 2 * PLANE) = make_uint2(l0, l1);
                    }
                }
        }
#endif
        if (g < 32) {
            write_w(0);
            const int gn = next_bit(gm, g);
            if (gn < 32) load_w(gn);
        }
        TS_STAMP(3 + ps * 5);
        ts_barrier();
        TS_STAMP(4 + ps * 5);
        // ---- offsets of this pass, software-pipelined: while offset kk is multiplied, the fragments of the next offset with work
        // in the pass (its loc entries were read one offset earlier) are already on their way from the halo image ----
        int step = 0;
        struct Ops { ts_bf16x8 x[RT][CS32][3]; bool any[RT]; };
        auto read_lc = [&](int kk, int (&lcv)[RT]) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) lcv[rt] = *reinterpret_cast<const unsigned short*>(ll + kk * (T * 2) + loc_lane + rt * 128);
        };
        auto fetch = [&](Ops& o, const int (&lcv)[RT]) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int s0 = lcv[rt] - ps * H;
                const bool ok = (unsigned)s0 < (unsigned)cnt;
                o.any[rt] = __ballot(ok) != 0ull;
                {   // unconditional (a sub-tile without work reads the zero row): reads inside a branch leave the number of outstanding
                    // LDS operations unknown at the join and the compiler waits for lgkmcnt(0) -- i.e. for THESE reads -- in front of
                    // the current offset's MFMAs (seen in spconv_rs_k: 9.5 k -> 8.0 k cycles per tile)
                    const int a0 = ts_slot_off(ok ? s0 : H, q);
#pragma unroll
                    for (int u = 0; u < CS32; ++u)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            if (U3D_TS_ABL & 32) o.x[rt][u][pl] = __builtin_bit_cast(ts_bf16x8, f32x4{(float)a0, 1.f, 2.f, (float)pl});
                            else o.x[rt][u][pl] = *reinterpret_cast<const ts_bf16x8*>(hl + (u * 3 + pl) * PLANE + a0);
                        }
                }
            }
        };
        // one offset: kk multiplied from `cur`; kn (32: none) fetched into `nxt` from its loc entries `lc_kn`; loc entries of kn2 -> lc_dst
        auto body = [&](int kk, Ops& cur, int kn, Ops& nxt, const int (&lc_kn)[RT], int kn2, int (&lc_dst)[RT]) {
            const int gk = kk / KG;
            const int jj = p.flip ? KG - 1 - (kk - gk * KG) : kk - gk * KG;      // the offset's weights inside the step's block
            const char* const wsl = wl + (NSLOT == 2 ? (step & 1) * WSLOT : 0) + w_rd;
            auto wfrag = [&](int u, int nb, f32x4 (&wf)[3]) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    if (U3D_TS_ABL & 32) wf[pl] = f32x4{(float)lane, 1.f, (float)nb, (float)pl};
                    else wf[pl] = *reinterpret_cast<const f32x4*>(wsl + ((((((nb >> 1) * KG + jj) * CS32 + u) * 2 + (nb & 1)) * 3 + pl) * 1024));
                }
            };
            f32x4 wf0[3];
            wfrag(0, 0, wf0);                                      // the first weight fragments are asked for before the next offset's rows
            if (kn < 32) {
                fetch(nxt, lc_kn);
                if (kn2 < 32) read_lc(kn2, lc_dst);
            }
#pragma unroll
            for (int u = 0; u < CS32; ++u)
#pragma unroll
                for (int nb = 0; nb < CD16; ++nb) {
                    f32x4 wf[3];
                    if (u == 0 && nb == 0) { wf[0] = wf0[0]; wf[1] = wf0[1]; wf[2] = wf0[2]; }
                    else wfrag(u, nb, wf);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        if (cur.any[rt]) {
                            f32x4 c = lo[rt][nb];
                            c = ts_mfma(wf[0], cur.x[rt][u][2], c);
                            c = ts_mfma(wf[1], cur.x[rt][u][1], c);
                            c = ts_mfma(wf[2], cur.x[rt][u][0], c);
                            c = ts_mfma(wf[0], cur.x[rt][u][1], c);
                            c = ts_mfma(wf[1], cur.x[rt][u][0], c);
                            lo[rt][nb] = c;
                            hi[rt][nb] = ts_mfma(wf[0], cur.x[rt][u][0], hi[rt][nb]);
                        }
                }
            if (kn < 32 && kn / KG != gk) {                        // hand over to the next step of this pass
                const int gn = kn / KG;
                if constexpr (NSLOT == 1 && !(U3D_TS_ABL & 8)) ts_barrier();           // every wave has read the slot for the last time
                write_w(NSLOT == 2 ? ((step + 1) & 1) : 0);
                const int gn2 = next_bit(gm, gn);
                if (gn2 < 32) load_w(gn2);
                if (!(U3D_TS_ABL & 8)) ts_barrier();
                ++step;
            }
        };
        if (g < 32 && !(U3D_TS_ABL & 128)) {
            int kk = __builtin_ctz(km), kn = next_bit(km, kk);
            int lcA[RT], lcB[RT];
            Ops A, B;
            read_lc(kk, lcA);
            if (kn < 32) read_lc(kn, lcB);
            fetch(A, lcA);
            while (true) {
                int kn2 = kn < 32 ? next_bit(km, kn) : 32;
                body(kk, A, kn, B, lcB, kn2, lcA);
                if (kn >= 32) break;
                kk = kn; kn = kn2;
                kn2 = kn < 32 ? next_bit(km, kn) : 32;
                body(kk, B, kn, A, lcA, kn2, lcB);
                if (kn >= 32) break;
                kk = kn; kn = kn2;
            }
        }
        TS_STAMP(5 + ps * 5);
#ifdef U3D_TS_TRACE
        if (tr && tid == 0 && ps < 5) tr[26 + ps] = (unsigned long long)__builtin_popcount(km);
#endif
    }
    TS_STAMP(31);

    // ---- write-out: lane (i16, q) holds columns 16 nb + 4 q .. + 3 of row i16 of each sub-tile ----
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int64_t row = r0 + rt * 64 + wave * 16 + i16;
        if (row < p.n) {
#pragma unroll
            for (int nb = 0; nb < CD16; ++nb) {
                f32x4 v = hi[rt][nb] + lo[rt][nb];
                const int64_t o = row * p.Cd + nb * 16 + q * 4;
                if (p.addend) v += *reinterpret_cast<const f32x4*>(p.addend + o);
                if ((U3D_TS_ABL & 64) && v[0] != 12345.f) continue;      // ablation: (almost) no write-out
                *reinterpret_cast<f32x4*>(p.out + o) = v;
            }
        }
    }
}

template <int CS32, int CD16, int RT, int H, int KG>
static int launch_ts(const TsParams& p, hipStream_t s) {
    constexpr int NSLOT = ts_nslot(CS32, CD16, RT, H, KG);
    constexpr int lds = ts_lds_bytes(CS32, CD16, RT, H, KG, NSLOT);
    static_assert(lds <= 160 * 1024, "tile exceeds the LDS");
    // the attribute is per device and cheap to set: no cache keyed by ordinal (a 64-entry table aliased devices >= 64)
    if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(&spconv_ts_k<CS32, CD16, RT, H, KG, NSLOT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((spconv_ts_k<CS32, CD16, RT, H, KG, NSLOT>), dim3((unsigned)p.n_tiles), dim3(256), lds, s, p);
    return check_launch("spconv_ts");
}

// ---- register-stationary weights (round 6, second form) ---------------------------------------------------------------------------
// What the cycle trace of spconv_ts_k says (profiles/round6_ts_cycle_trace.txt): a step -- one offset of one pass -- lasts 1100-1600
// cycles for ~220 cycles of MFMA issue per wave; the four waves meet at a barrier per offset (the weight hand-over) with unequal work,
// and the weights of an offset (6 KB) are fetched again for every (tile, pass, offset): 786 MB per level-1 launch through the texture
// path, more than the feature rows.  For a 32 -> 32 block the weights of all 27 offsets are 162 KB in three planes -- a third of a CU's
// register file.  So here they never move: a PERSISTENT workgroup (one per CU, one wave per SIMD, up to 512 registers per lane) loads
// them once, wave w keeping the MFMA fragments of "its" 6-7 offsets (one centre / face / edge / corner mix per wave) in registers for
// the whole launch, and walks a contiguous range of 64-row tiles.  Per tile: the halo rows are split into their planes once (as in
// spconv_ts_k), then every wave multiplies ITS offsets for ALL four 16-row sub-tiles -- no weight traffic, no barrier until the tile is
// done -- and the four partial [64 x 32] tiles are added in wave order through LDS (fixed order: deterministic).  Wider layers are
// tiled over 32-channel blocks of source and destination (Cs/32 x Cd/32 launches inside the entry point, accumulating over the source
// blocks through `addend`).
constexpr int RS_T = 64;
__device__ const signed char RS_OFFS[4][7] = {{13, 4, 1, 3, 5, 0, 2}, {10, 12, 7, 9, 11, 6, 8}, {14, 16, 15, 17, 19, 18, 20}, {22, 21, 23, 25, 24, 26, -1}};

struct RsParams {
    const float* src;
    const void* w;
    const int32_t* nhalo;
    const int32_t* halo;
    const uint16_t* loc;
    const float* addend;       // nullable; same leading dimension and column offset as out
    float* out;
    int64_t n;
    int src_ld, src_c0;        // floats per source row, first channel of this launch's 32-channel block
    int dst_ld, dst_c0;
    int w_base16;              // 16-byte index of (slice, offset 0, unit) in the packed weights; w_k16: stride between offsets
    int w_k16;
    int n_tiles, tiles_per_wg, flip;
    unsigned long long* trace; // -DU3D_TS_TRACE builds: [workgroups][4 waves][8] accumulated cycles per phase
};

constexpr int rs_lds_bytes(int h) { return 3 * (h + 1) * 64 + 27 * RS_T * 2 + 4 * RS_T * 32 * 4; }

template <int H>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void spconv_rs_k(RsParams p) {
    extern __shared__ __attribute__((aligned(16))) char ts_smem[];
    constexpr int T = RS_T, NC = 27 * T, PLANE = (H + 1) * 64, HALO = 3 * PLANE, LOCB = 27 * T * 2, ROUNDS = H / 32;
    char* const hl = ts_smem;
    char* const ll = ts_smem + HALO;
    float* const red = reinterpret_cast<float*>(ts_smem + HALO + LOCB);          // [4 waves][64 rows][32 columns]
    const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = (int)xcd_swizzle(blockIdx.x, gridDim.x);                      // neighbouring tile ranges on one XCD / L2
    const int t_begin = wg * p.tiles_per_wg, t_end = min(p.n_tiles, t_begin + p.tiles_per_wg);
    const __amdgpu_buffer_rsrc_t rs_src = make_rsrc(p.src, p.n * p.src_ld * 4);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w);
    const int ld4 = p.src_ld * 4, c0b = p.src_c0 * 4;
    const int pc = tid & 7, prow = tid >> 3;
#ifdef U3D_TS_TRACE
    unsigned long long* const tr = p.trace ? p.trace + ((int64_t)blockIdx.x * 4 + wave) * 8 : nullptr;
    unsigned long long tprev = clock64();
#define RS_PHASE(i) do { if (tr && lane == 0) { const unsigned long long now_ = clock64(); tr[(i)] += now_ - tprev; tprev = now_; } } while (0)
#else
#define RS_PHASE(i) do {} while (0)
#endif

    // ---- this wave's weights: fragments of its offsets, in registers for the whole launch ----
    f32x4 wf[7][2][3];
    int kks[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int kk = RS_OFFS[wave][j];
        kks[j] = kk;
        const int k = kk < 0 ? 0 : (p.flip ? 26 - kk : kk);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wf[j][nb][pl] = bload128(rs_w, (p.w_base16 + k * p.w_k16 + (nb * 3 + pl) * 64 + lane) * 16, 0);
    }
    if (tid < 12) *reinterpret_cast<f32x4*>(hl + (tid >> 2) * PLANE + H * 64 + (tid & 3) * 16) = f32x4{0.f, 0.f, 0.f, 0.f};      // the zero row
    RS_PHASE(0);

    for (int tile = t_begin; tile < t_end; ++tile) {
        const int64_t r0 = (int64_t)tile * T;
        const __amdgpu_buffer_rsrc_t rs_h = make_rsrc(p.halo + (int64_t)tile * NC, (int64_t)NC * 4);
        const __amdgpu_buffer_rsrc_t rs_l = make_rsrc(p.loc + (int64_t)tile * NC, (int64_t)NC * 2);
        const int nh = __builtin_amdgcn_readfirstlane(p.nhalo[tile]);
        const int npass = (nh + H - 1) / H;
        f32x4 hi[4][2], lo[4][2];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) { hi[rt][nb] = f32x4{0.f, 0.f, 0.f, 0.f}; lo[rt][nb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int ps = 0; ps < npass; ++ps) {
            const int cnt = min(H, nh - ps * H);
            ts_barrier();                                   // the previous tile's (pass's) reads of the halo image, loc table and `red` are done
            RS_PHASE(1);
            if (ps == 0 && tid < LOCB / 16) *reinterpret_cast<f32x4*>(ll + tid * 16) = bload128(rs_l, tid * 16, 0);
            int hrow[ROUNDS];
#pragma unroll
            for (int j = 0; j < ROUNDS; ++j) hrow[j] = bload32(rs_h, (ps * H + j * 32 + prow) * 4, 0);
#pragma unroll
            for (int j0 = 0; j0 < ROUNDS; j0 += 5) {
                f32x4 v[5];
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    if (j0 + j < ROUNDS) {
                        const bool ok = (j0 + j) * 32 + prow < cnt;
                        v[j] = bload128(rs_src, ok ? (int)__umul24(hrow[j0 + j], ld4) + c0b + pc * 16 : 0x7ffffff0, 0);
                    }
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    if (j0 + j < ROUNDS) {
                        const int sl = (j0 + j) * 32 + prow;
                        unsigned h0, m0, l0, h1, m1, l1;
                        split3_pair(v[j][0], v[j][1], h0, m0, l0);
                        split3_pair(v[j][2], v[j][3], h1, m1, l1);
                        char* const b = hl + ts_slot_off(sl, pc & 3) + (pc >> 2) * 8;
                        *reinterpret_cast<uint2*>(b) = make_uint2(h0, h1);
                        *reinterpret_cast<uint2*>(b + PLANE) = make_uint2(m0, m1);
                        *reinterpret_cast<uint2*>(b + 2 * PLANE) = make_uint2(l0, l1);
                    }
            }
            RS_PHASE(2);
            ts_barrier();
            RS_PHASE(3);
            // ---- this wave's 28 items (offset x sub-tile): all positions first (one LDS round trip for the lot), then per item three
            // fragment reads one item ahead of the twelve MFMAs ----
            int a0s[28];
            unsigned amask = 0u;
#pragma unroll
            for (int it = 0; it < 28; ++it) {
                const int kk = kks[it >> 2];
                const int l = kk < 0 ? 0xFFFF : (int)*reinterpret_cast<const unsigned short*>(ll + kk * (T * 2) + ((it & 3) * 16 + i16) * 2);
                const int s0 = l - ps * H;
                const bool ok = (unsigned)s0 < (unsigned)cnt;
                a0s[it] = ts_slot_off(ok ? s0 : H, q);
                amask |= (__ballot(ok) != 0ull ? 1u : 0u) << it;
            }
            // (the fragment reads are UNCONDITIONAL -- an item without work reads the zero row: a read inside a branch makes the number
            // of outstanding LDS operations unknown at the join, and the compiler then waits for lgkmcnt(0) in front of every item's
            // MFMAs, i.e. for the reads of the NEXT item as well: no overlap at all, measured 9.5 k cycles per tile instead of ~5 k)
            ts_bf16x8 xa[3], xb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) xa[pl] = *reinterpret_cast<const ts_bf16x8*>(hl + pl * PLANE + a0s[0]);
#pragma unroll
            for (int it = 0; it < 28; ++it) {
                const int j = it >> 2, rt = it & 3;
                ts_bf16x8 (&cur)[3] = (it & 1) ? xb : xa;
                ts_bf16x8 (&nxt)[3] = (it & 1) ? xa : xb;
                if (it + 1 < 28) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) nxt[pl] = *reinterpret_cast<const ts_bf16x8*>(hl + pl * PLANE + a0s[it + 1]);
                }
                if ((amask >> it) & 1u) {
                    // the two column blocks' chains alternate: a dependent MFMA waits for its predecessor (one wave per SIMD: nobody else fills the gap)
                    f32x4 c0 = lo[rt][0], c1 = lo[rt][1];
                    c0 = ts_mfma(wf[j][0][0], cur[2], c0); c1 = ts_mfma(wf[j][1][0], cur[2], c1);
                    hi[rt][0] = ts_mfma(wf[j][0][0], cur[0], hi[rt][0]); hi[rt][1] = ts_mfma(wf[j][1][0], cur[0], hi[rt][1]);
                    c0 = ts_mfma(wf[j][0][1], cur[1], c0); c1 = ts_mfma(wf[j][1][1], cur[1], c1);
                    c0 = ts_mfma(wf[j][0][2], cur[0], c0); c1 = ts_mfma(wf[j][1][2], cur[0], c1);
                    c0 = ts_mfma(wf[j][0][0], cur[1], c0); c1 = ts_mfma(wf[j][1][0], cur[1], c1);
                    c0 = ts_mfma(wf[j][0][1], cur[0], c0); c1 = ts_mfma(wf[j][1][1], cur[0], c1);
                    lo[rt][0] = c0; lo[rt][1] = c1;
                }
            }
            RS_PHASE(4);
        }
        // ---- the four waves' partial tiles -> LDS -> summed in wave order by the wave that owns the rows -> dst ----
        if (npass == 0) ts_barrier();
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                *reinterpret_cast<f32x4*>(red + ((wave * T + rt * 16 + i16) * 32 + nb * 16 + q * 4)) = hi[rt][nb] + lo[rt][nb];
        ts_barrier();
        RS_PHASE(5);
        {
            const int rr = wave * 16 + i16;                 // this lane's row of the tile
            const int64_t row = r0 + rr;
            if (row < p.n) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(red + ((0 * T + rr) * 32 + nb * 16 + q * 4));
#pragma unroll
                    for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(red + ((w * T + rr) * 32 + nb * 16 + q * 4));
                    const int64_t o = row * p.dst_ld + p.dst_c0 + nb * 16 + q * 4;
                    if (p.addend) v += *reinterpret_cast<const f32x4*>(p.addend + o);
                    *reinterpret_cast<f32x4*>(p.out + o) = v;
                }
            }
        }
        RS_PHASE(6);
    }
}

template <int H>
static int launch_rs(const RsParams& p, int wgs, hipStream_t s) {
    constexpr int lds = rs_lds_bytes(H);
    static_assert(lds <= 160 * 1024, "tile exceeds the LDS");
    hipFuncSetAttribute(reinterpret_cast<const void*>(&spconv_rs_k<H>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((spconv_rs_k<H>), dim3((unsigned)wgs), dim3(256), lds, s, p);
    return check_launch("spconv_rs");
}

// ---- the same, bf16 operands from bf16 ROWS (BASELINE configs[2]; the shadows the batch-norm kernels write, bn.hip store_shadow) ----
// One plane instead of three: a wave's 6-7 offsets are 56 registers, a halo row is 64 bytes that arrive in MFMA fragment order (no
// split, no conversion: a 16-byte copy per lane), an item is two MFMAs.  The kernel fits three to four workgroups per CU (the partial
// tiles of the final reduction reuse the halo image's LDS), so one workgroup's load phase runs under the others' arithmetic -- what
// the three-plane form cannot do with 168 weight registers per wave.
struct RsbParams {
    const void* src;           // bf16 [n][src_ld] in fragment order per 32-channel group
    const void* w;             // u3d_weight_pack_bf16 layout
    const int32_t* nhalo;
    const int32_t* halo;
    const uint16_t* loc;
    const float* addend;
    float* out;
    int64_t n;
    int src_ld, src_c0, dst_ld, dst_c0, w_base16, w_k16;
    int n_tiles, tiles_per_wg, flip;
};

constexpr int rsb_lds_bytes(int h) {
    return ((h + 1) * 64 + 27 * RS_T * 2) > 4 * RS_T * 32 * 4 ? ((h + 1) * 64 + 27 * RS_T * 2) : 4 * RS_T * 32 * 4;
}

template <int H>
__global__ __launch_bounds__(256) void spconv_rsb_k(RsbParams p) {
    extern __shared__ __attribute__((aligned(16))) char ts_smem[];
    constexpr int T = RS_T, NC = 27 * T, HALO = (H + 1) * 64, LOCB = 27 * T * 2, ROUNDS = H / 64;
    static_assert(H % 64 == 0, "halo rows per pass");
    char* const hl = ts_smem;
    char* const ll = ts_smem + HALO;
    float* const red = reinterpret_cast<float*>(ts_smem);                        // aliases the halo image and the loc table (barrier in between)
    const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = (int)xcd_swizzle(blockIdx.x, gridDim.x);
    const int t_begin = wg * p.tiles_per_wg, t_end = min(p.n_tiles, t_begin + p.tiles_per_wg);
    const __amdgpu_buffer_rsrc_t rs_src = make_rsrc(p.src, p.n * p.src_ld * 2);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w);
    const int ld2 = p.src_ld * 2, c0b = (p.src_c0 >> 5) * 64;
    const int ch = tid & 3, prow = tid >> 2;                   // load phase: 16-byte chunk of the row's 64, row within a round of 64

    f32x4 wf[7][2];
    int kks[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int kk = RS_OFFS[wave][j];
        kks[j] = kk;
        const int k = kk < 0 ? 0 : (p.flip ? 26 - kk : kk);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) wf[j][nb] = bload128(rs_w, (p.w_base16 + k * p.w_k16 + nb * 64 + lane) * 16, 0);
    }

    for (int tile = t_begin; tile < t_end; ++tile) {
        const int64_t r0 = (int64_t)tile * T;
        const __amdgpu_buffer_rsrc_t rs_h = make_rsrc(p.halo + (int64_t)tile * NC, (int64_t)NC * 4);
        const __amdgpu_buffer_rsrc_t rs_l = make_rsrc(p.loc + (int64_t)tile * NC, (int64_t)NC * 2);
        const int nh = __builtin_amdgcn_readfirstlane(p.nhalo[tile]);
        const int npass = (nh + H - 1) / H;
        f32x4 acc[4][2];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) acc[rt][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ps = 0; ps < npass; ++ps) {
            const int cnt = min(H, nh - ps * H);
            ts_barrier();                                   // the previous tile's sums are out of `red`, the previous pass's reads done
            if (tid < 4) *reinterpret_cast<f32x4*>(hl + H * 64 + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};      // the zero row (red overwrote it)
            if (tid < LOCB / 16) *reinterpret_cast<f32x4*>(ll + tid * 16) = bload128(rs_l, tid * 16, 0);
            int hrow[ROUNDS];
#pragma unroll
            for (int j = 0; j < ROUNDS; ++j) hrow[j] = bload32(rs_h, (ps * H + j * 64 + prow) * 4, 0);
            f32x4 v[ROUNDS];
#pragma unroll
            for (int j = 0; j < ROUNDS; ++j) {
                const bool ok = j * 64 + prow < cnt;
                v[j] = bload128(rs_src, ok ? (int)__umul24(hrow[j], ld2) + c0b + ch * 16 : 0x7ffffff0, 0);
            }
#pragma unroll
            for (int j = 0; j < ROUNDS; ++j) *reinterpret_cast<f32x4*>(hl + ts_slot_off(j * 64 + prow, ch)) = v[j];
            ts_barrier();
            int a0s[28];
            unsigned amask = 0u;
#pragma unroll
            for (int it = 0; it < 28; ++it) {
                const int kk = kks[it >> 2];
                const int l = kk < 0 ? 0xFFFF : (int)*reinterpret_cast<const unsigned short*>(ll + kk * (T * 2) + ((it & 3) * 16 + i16) * 2);
                const int s0 = l - ps * H;
                const bool ok = (unsigned)s0 < (unsigned)cnt;
                a0s[it] = ts_slot_off(ok ? s0 : H, q);
                amask |= (__ballot(ok) != 0ull ? 1u : 0u) << it;
            }
            ts_bf16x8 xa, xb;
            xa = *reinterpret_cast<const ts_bf16x8*>(hl + a0s[0]);
#pragma unroll
            for (int it = 0; it < 28; ++it) {
                const int j = it >> 2, rt = it & 3;
                ts_bf16x8& cur = (it & 1) ? xb : xa;
                ts_bf16x8& nxt = (it & 1) ? xa : xb;
                if (it + 1 < 28) nxt = *reinterpret_cast<const ts_bf16x8*>(hl + a0s[it + 1]);      // unconditional: see spconv_rs_k
                if ((amask >> it) & 1u) {
                    acc[rt][0] = ts_mfma(wf[j][0], cur, acc[rt][0]);
                    acc[rt][1] = ts_mfma(wf[j][1], cur, acc[rt][1]);
                }
            }
        }
        ts_barrier();                                       // every wave is done with the halo image and the loc table: `red` may overwrite them
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                *reinterpret_cast<f32x4*>(red + ((wave * T + rt * 16 + i16) * 32 + nb * 16 + q * 4)) = acc[rt][nb];
        ts_barrier();
        {
            const int rr = wave * 16 + i16;
            const int64_t row = r0 + rr;
            if (row < p.n) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(red + ((0 * T + rr) * 32 + nb * 16 + q * 4));
#pragma unroll
                    for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(red + ((w * T + rr) * 32 + nb * 16 + q * 4));
                    const int64_t o = row * p.dst_ld + p.dst_c0 + nb * 16 + q * 4;
                    if (p.addend) v += *reinterpret_cast<const f32x4*>(p.addend + o);
                    *reinterpret_cast<f32x4*>(p.out + o) = v;
                }
            }
        }
    }
}

template <int H>
static int launch_rsb(const RsbParams& p, int wgs, hipStream_t s) {
    constexpr int lds = rsb_lds_bytes(H);
    hipLaunchKernelGGL((spconv_rsb_k<H>), dim3((unsigned)wgs), dim3(256), lds, s, p);
    return check_launch("spconv_rsb");
}

// launch plan: rows per tile T and halo rows per pass H by shape (tools/prof_ts.py sweeps; U3D_TS_T / U3D_TS_H override for A/B runs)
static bool ts_plan(int Cs, int Cd, int64_t n, int* T, int* H) {
    if (Cs % 32 || Cd % 32 || n <= 0) return false;
    const int cs32 = Cs / 32, cd16 = Cd / 16;
    int t = 0, h = 0;
    if (cs32 == 1 && cd16 == 2) { t = 128; h = 192; }
    else if (cs32 == 2 && cd16 == 2) { t = 128; h = 128; }
    else if (cs32 == 1 && cd16 == 4) { t = 128; h = 192; }
    else if (cs32 == 2 && cd16 == 4) { t = 128; h = 128; }
    else return false;
    const char* et = getenv("U3D_TS_T");      // read per call: tools/prof_ts.py sweeps the plan inside one process
    const char* eh = getenv("U3D_TS_H");
    if (et && atoi(et) > 0) t = atoi(et);
    if (eh && atoi(eh) > 0) h = atoi(eh);
    *T = t; *H = h;
    return true;
}

}  // namespace u3d

using namespace u3d;

// compute units of the current device (one attribute query, no property struct: this sits on the launch path)
static int ts_cu_count() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
}

extern "C" {

int u3d_spconv_ts_plan(int Cs, int Cd, int64_t n, int* tile_rows, int* halo_rows) {
    if (!tile_rows || !halo_rows || !ts_plan(Cs, Cd, n, tile_rows, halo_rows)) return U3D_EUNSUPPORTED;
    return U3D_OK;
}

int u3d_subm_halo_pmax(int tile_rows, int halo_rows) {
    if (tile_rows <= 0 || halo_rows <= 0) return U3D_EINVAL;
    return (27 * tile_rows + halo_rows - 1) / halo_rows;
}

int u3d_subm_halo(const int32_t* coords, int64_t n, const uint64_t* bitmap, const int32_t* word_rank, int64_t hash_slots, int B,
                  int X, int Y, int Z, int tile_rows, int halo_rows, int32_t* nhalo, int32_t* halo, uint16_t* loc, uint32_t* pmask,
                  u3d_stream_t stream) {
    if (!coords || !bitmap || !word_rank || !nhalo || !halo || !loc || !pmask || n <= 0 || halo_rows <= 0) return U3D_EINVAL;
    if (n >= (1 << 24)) { set_error("subm_halo: %lld rows exceed the kernels' 24-bit row indices", (long long)n); return U3D_EUNSUPPORTED; }
    const int pmax = u3d_subm_halo_pmax(tile_rows, halo_rows);
    if (pmax > 64) { set_error("subm_halo: %d passes per tile (tile %d, halo %d) exceed 64", pmax, tile_rows, halo_rows); return U3D_EUNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_RULEBOOK, s, 0.0);
    const Index ix = make_index(bitmap, word_rank, B, X, Y, Z, hash_slots);
    const unsigned nt = (unsigned)ceil_div(n, tile_rows);
    if (tile_rows == 64) hipLaunchKernelGGL((subm_halo_k<64, 2048>), dim3(nt), dim3(256), 0, s, coords, n, ix, halo_rows, pmax, nhalo, halo, loc, pmask);
    else if (tile_rows == 128) hipLaunchKernelGGL((subm_halo_k<128, 4096>), dim3(nt), dim3(256), 0, s, coords, n, ix, halo_rows, pmax, nhalo, halo, loc, pmask);
    else if (tile_rows == 256) hipLaunchKernelGGL((subm_halo_k<256, 8192>), dim3(nt), dim3(256), 0, s, coords, n, ix, halo_rows, pmax, nhalo, halo, loc, pmask);
    else { set_error("subm_halo: tile_rows %d not in {64, 128, 256}", tile_rows); return U3D_EUNSUPPORTED; }
    return check_launch("subm_halo");
}

int u3d_spconv_ts_x3(const float* src, int64_t n, const void* w_rows_x3, const int32_t* nhalo, const int32_t* halo, const uint16_t* loc,
                     const uint32_t* pmask, int tile_rows, int halo_rows, int flip, int Cs, int Cd, const float* addend, float* dst,
                     double flops_hint, u3d_stream_t stream) {
    if (!src || !w_rows_x3 || !nhalo || !halo || !loc || !pmask || !dst || n <= 0) return U3D_EINVAL;
    if (n >= (1 << 24) || n * Cs * 4 >= 0x7fffffffLL) {
        set_error("spconv_ts: %lld rows x %d channels exceed the kernel's 32-bit addressing", (long long)n, Cs);
        return U3D_EUNSUPPORTED;
    }
    if (Cs % 32 || Cd % 32) { set_error("spconv_ts: Cs=%d Cd=%d must be multiples of 32", Cs, Cd); return U3D_EUNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_CONV_FWD, s, flops_hint);
    TsParams p;
    p.src = src; p.w = w_rows_x3; p.nhalo = nhalo; p.halo = halo; p.loc = loc; p.pmask = pmask; p.addend = addend; p.out = dst;
    p.trace = nullptr;
#ifdef U3D_TS_TRACE
    { const char* e = getenv("U3D_TS_TRACE_PTR"); if (e) p.trace = (unsigned long long*)strtoull(e, nullptr, 0); }
#endif
    p.n = n; p.Cs = Cs; p.Cd = Cd; p.n_tiles = (int)ceil_div(n, tile_rows); p.pmax = u3d_subm_halo_pmax(tile_rows, halo_rows); p.flip = flip ? 1 : 0;
    const int cs32 = Cs / 32, cd16 = Cd / 16;
    const char* ekg = getenv("U3D_TS_KG");
    const int kg = ekg && atoi(ekg) == 1 ? 1 : 3;
#define U3D_TS_CASE(cs, cd, rt, h) if (cs32 == cs && cd16 == cd && tile_rows == 64 * rt && halo_rows == h) return kg == 3 ? launch_ts<cs, cd, rt, h, 3>(p, s) : launch_ts<cs, cd, rt, h, 1>(p, s);
    U3D_TS_CASE(1, 2, 2, 128) U3D_TS_CASE(1, 2, 2, 192) U3D_TS_CASE(1, 2, 2, 256)
    U3D_TS_CASE(1, 2, 4, 128) U3D_TS_CASE(1, 2, 4, 192) U3D_TS_CASE(1, 2, 4, 256)
    U3D_TS_CASE(1, 2, 1, 128)
    U3D_TS_CASE(2, 2, 2, 128) U3D_TS_CASE(2, 2, 2, 192) U3D_TS_CASE(2, 2, 4, 128)
    U3D_TS_CASE(1, 4, 2, 128) U3D_TS_CASE(1, 4, 2, 192) U3D_TS_CASE(1, 4, 4, 128)
    U3D_TS_CASE(2, 4, 2, 128) U3D_TS_CASE(2, 4, 1, 128) U3D_TS_CASE(2, 4, 2, 192) U3D_TS_CASE(2, 4, 4, 128)
#undef U3D_TS_CASE
    set_error("spconv_ts: no instantiation for Cs=%d Cd=%d tile %d halo %d", Cs, Cd, tile_rows, halo_rows);
    return U3D_EUNSUPPORTED;
}

int u3d_spconv_rs_x3(const float* src, int64_t n, const void* w_rows_x3, const int32_t* nhalo, const int32_t* halo, const uint16_t* loc,
                     int halo_rows, int flip, int Cs, int Cd, const float* addend, float* dst, int workgroups, double flops_hint, u3d_stream_t stream) {
    if (!src || !w_rows_x3 || !nhalo || !halo || !loc || !dst || n <= 0) return U3D_EINVAL;
    if (n >= (1 << 24) || n * Cs * 4 >= 0x7fffffffLL) {
        set_error("spconv_rs: %lld rows x %d channels exceed the kernel's 32-bit addressing", (long long)n, Cs);
        return U3D_EUNSUPPORTED;
    }
    if (Cs % 32 || Cd % 32 || Cs > 256 || Cd > 256) { set_error("spconv_rs: Cs=%d Cd=%d must be multiples of 32 up to 256", Cs, Cd); return U3D_EUNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_CONV_FWD, s, flops_hint);
    if (workgroups <= 0) workgroups = ts_cu_count();
    RsParams p;
    p.src = src; p.w = w_rows_x3; p.nhalo = nhalo; p.halo = halo; p.loc = loc; p.out = dst; p.n = n;
    p.src_ld = Cs; p.dst_ld = Cd; p.flip = flip ? 1 : 0;
    p.n_tiles = (int)ceil_div(n, RS_T);
    p.tiles_per_wg = (int)ceil_div(p.n_tiles, workgroups);
    const int wgs = (int)ceil_div(p.n_tiles, p.tiles_per_wg);
    const int cs32 = Cs / 32;
    p.w_k16 = cs32 * 384;
    p.trace = nullptr;
#ifdef U3D_TS_TRACE
    { const char* e = getenv("U3D_TS_TRACE_PTR"); if (e) p.trace = (unsigned long long*)strtoull(e, nullptr, 0); }
#endif
    // destination blocks outer, source blocks inner: block (ds, ss) adds src[:, 32 ss ..] . W into dst[:, 32 ds ..] on top of what the
    // previous source block left there (the first one on top of `addend`)
    for (int ds = 0; ds < Cd / 32; ++ds)
        for (int ss = 0; ss < cs32; ++ss) {
            p.src_c0 = ss * 32; p.dst_c0 = ds * 32;
            p.w_base16 = (ds * 27 * cs32 + ss) * 384;
            p.addend = ss == 0 ? addend : dst;
            int rc = U3D_EUNSUPPORTED;
            if (halo_rows == 256) rc = launch_rs<256>(p, wgs, s);
            else if (halo_rows == 320) rc = launch_rs<320>(p, wgs, s);
            else if (halo_rows == 416) rc = launch_rs<416>(p, wgs, s);
            else set_error("spconv_rs: halo_rows %d not in {256, 320, 416}", halo_rows);
            if (rc != U3D_OK) return rc;
        }
    return U3D_OK;
}

int u3d_spconv_rs_bf16a(const void* src_bf16, int64_t n, const void* w_rows_bf16, const int32_t* nhalo, const int32_t* halo, const uint16_t* loc,
                        int halo_rows, int flip, int Cs, int Cd, const float* addend, float* dst, int workgroups, double flops_hint, u3d_stream_t stream) {
    if (!src_bf16 || !w_rows_bf16 || !nhalo || !halo || !loc || !dst || n <= 0) return U3D_EINVAL;
    if (n >= (1 << 24) || n * Cs * 2 >= 0x7fffffffLL) {
        set_error("spconv_rs_bf16a: %lld rows x %d channels exceed the kernel's 32-bit addressing", (long long)n, Cs);
        return U3D_EUNSUPPORTED;
    }
    if (Cs % 32 || Cd % 32 || Cs > 256 || Cd > 256) { set_error("spconv_rs_bf16a: Cs=%d Cd=%d must be multiples of 32 up to 256", Cs, Cd); return U3D_EUNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_CONV_FWD, s, flops_hint);
    if (workgroups <= 0) workgroups = 3 * ts_cu_count();
    RsbParams p;
    p.src = src_bf16; p.w = w_rows_bf16; p.nhalo = nhalo; p.halo = halo; p.loc = loc; p.out = dst; p.n = n;
    p.src_ld = Cs; p.dst_ld = Cd; p.flip = flip ? 1 : 0;
    p.n_tiles = (int)ceil_div(n, RS_T);
    p.tiles_per_wg = (int)ceil_div(p.n_tiles, workgroups);
    const int wgs = (int)ceil_div(p.n_tiles, p.tiles_per_wg);
    const int cs32 = Cs / 32;
    p.w_k16 = cs32 * 128;
    for (int ds = 0; ds < Cd / 32; ++ds)
        for (int ss = 0; ss < cs32; ++ss) {
            p.src_c0 = ss * 32; p.dst_c0 = ds * 32;
            p.w_base16 = (ds * 27 * cs32 + ss) * 128;
            p.addend = ss == 0 ? addend : dst;
            int rc = U3D_EUNSUPPORTED;
            if (halo_rows == 256) rc = launch_rsb<256>(p, wgs, s);
            else if (halo_rows == 320) rc = launch_rsb<320>(p, wgs, s);
            else if (halo_rows == 448) rc = launch_rsb<448>(p, wgs, s);
            else set_error("spconv_rs_bf16a: halo_rows %d not in {256, 320, 448}", halo_rows);
            if (rc != U3D_OK) return rc;
        }
    return U3D_OK;
}

}  // extern "C"
