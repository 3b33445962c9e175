// R2 / R3 rulebooks on gfx950.
// Replaces spconv's indice-pair generation for SubMConv3d(k=3) and SparseConv3d(k=2,s=2)
// (reference call sites unidet3d/spconv_unet.py:43-56,148-154,178-183; unidet3d/unidet3d.py:97-103).
// Neighbour lookup goes through the occupancy bitmap + popcount rank (u3d_common.h) --
// the key -> row map is a direct-address table, so rows come out in canonical (sorted-key)
// order by construction and the pair lists are bit-exact against the CPU oracle.
// Pair lists are produced by a two-pass stable compaction: per-block counts (ballot/popcount),
// a per-offset scan of block counts, then a wave-64 ballot prefix inside each block.
#include "u3d_common.h"

namespace u3d {

constexpr int RB_T = 256;

// val[k*n + i] = parent row if offset(i) == k (and the parent is inside the halved grid) else -1
__global__ __launch_bounds__(RB_T) void down_nbr_k(const int32_t* __restrict__ coords, int64_t n, Index ix2, int32_t* val) {
    const int64_t i = (int64_t)blockIdx.x * RB_T + threadIdx.x;
    if (i >= n) return;
    const int4 c = *reinterpret_cast<const int4*>(coords + i * 4);
    const int k = ((c.y & 1) << 2) | ((c.z & 1) << 1) | (c.w & 1);
    const int prow = index_lookup(ix2, c.x, c.y >> 1, c.z >> 1, c.w >> 1);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) val[(int64_t)kk * n + i] = (kk == k) ? prow : -1;
}

__global__ __launch_bounds__(RB_T) void index_mark_k(const int32_t* __restrict__ coords, int64_t n, int shift, int X2, int Y2, int Z2,
                                                     int Zw2, unsigned long long* bitmap2) {
    const int64_t i = (int64_t)blockIdx.x * RB_T + threadIdx.x;
    if (i >= n) return;
    const int4 c = *reinterpret_cast<const int4*>(coords + i * 4);
    const int x = c.y >> shift, y = c.z >> shift, z = c.w >> shift;
    if ((unsigned)x >= (unsigned)X2 || (unsigned)y >= (unsigned)Y2 || (unsigned)z >= (unsigned)Z2) return;   // odd extent: edge voxel dropped
    const int64_t w = ((int64_t)(c.x * X2 + x) * Y2 + y) * Zw2 + (z >> 6);
    atomicOr(&bitmap2[w], 1ull << (z & 63));
}

// grid (nblk, K)
__global__ __launch_bounds__(RB_T) void rb_count_k(const int32_t* __restrict__ val, int64_t n, int nblk, int32_t* block_cnt) {
    const int64_t r = (int64_t)blockIdx.x * RB_T + threadIdx.x;
    const int k = blockIdx.y;
    const bool ok = r < n && val[(int64_t)k * n + r] >= 0;
    const int c = __syncthreads_count(ok);
    if (threadIdx.x == 0) block_cnt[(int64_t)k * nblk + blockIdx.x] = c;
}

// one block per offset: exclusive scan of its block counts
__global__ __launch_bounds__(RB_T) void rb_scan_k(const int32_t* __restrict__ block_cnt, int nblk, int32_t* block_base, int32_t* counts) {
    __shared__ int s_w[RB_T / 64];
    __shared__ int s_carry;
    const int k = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += RB_T) {
        const int i = base + threadIdx.x;
        const int v = i < nblk ? block_cnt[(int64_t)k * nblk + i] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) s_w[wave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < RB_T / 64; ++w) { int t = s_w[w]; if (w < wave) woff += t; tot += t; }
        const int carry = s_carry;
        if (i < nblk) block_base[(int64_t)k * nblk + i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) counts[k] = s_carry;
}

// stable compaction: list_row[k][pos] = r, list_val[k][pos] = val[k][r]   grid (nblk, K)
__global__ __launch_bounds__(RB_T) void rb_write_k(const int32_t* __restrict__ val, int64_t n, int nblk, const int32_t* __restrict__ block_base,
                                                   int64_t cap, int32_t* list_row, int32_t* list_val) {
    __shared__ int s_w[RB_T / 64];
    const int64_t r = (int64_t)blockIdx.x * RB_T + threadIdx.x;
    const int k = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = r < n ? val[(int64_t)k * n + r] : -1;
    const bool ok = v >= 0;
    const unsigned long long m = __ballot(ok);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int w = 0; w < RB_T / 64; ++w) if (w < wave) woff += s_w[w];
    if (ok) {
        const int64_t pos = (int64_t)k * cap + block_base[(int64_t)k * nblk + blockIdx.x] + woff + rank;
        list_row[pos] = (int)r;
        list_val[pos] = v;
    }
}

// ---- SubM rulebook without the dense neighbour matrix -------------------------------------------------------------------
// Rounds 1-2 stored nbr[27][n] (subm_nbr_k) and read it back in the count and the write pass: 3 x 27 n 4 B = 115 MB of traffic at
// level 1 of cfg2 for 34 MB of pairs.  The lookup is a bitmap word + a rank word (L2-resident), so both passes now redo it.
__device__ __forceinline__ int subm_lookup(const int32_t* __restrict__ coords, int64_t o, int k, const Index& ix) {
    const int dx = k / 9 - 1, dy = (k / 3) % 3 - 1, dz = k % 3 - 1;
    const int4 c = *reinterpret_cast<const int4*>(coords + o * 4);
    return index_lookup(ix, c.x, c.y + dx, c.z + dy, c.w + dz);
}

// grid (nblk, 27)
__global__ __launch_bounds__(RB_T) void subm_count_k(const int32_t* __restrict__ coords, int64_t n, Index ix, int nblk, int32_t* block_cnt) {
    const int64_t r = (int64_t)blockIdx.x * RB_T + threadIdx.x;
    const int k = blockIdx.y;
    const bool ok = r < n && subm_lookup(coords, r, k, ix) >= 0;
    const int c = __syncthreads_count(ok);
    if (threadIdx.x == 0) block_cnt[(int64_t)k * nblk + blockIdx.x] = c;
}

// stable compaction, as rb_write_k: list_row[k][pos] = output row, list_val[k][pos] = its neighbour's row    grid (nblk, 27)
__global__ __launch_bounds__(RB_T) void subm_write_k(const int32_t* __restrict__ coords, int64_t n, Index ix, int nblk,
                                                     const int32_t* __restrict__ block_base, int64_t cap, int32_t* list_row, int32_t* list_val) {
    __shared__ int s_w[RB_T / 64];
    const int64_t r = (int64_t)blockIdx.x * RB_T + threadIdx.x;
    const int k = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = r < n ? subm_lookup(coords, r, k, ix) : -1;
    const bool ok = v >= 0;
    const unsigned long long m = __ballot(ok);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int woff = 0;
#pragma unroll
    for (int w = 0; w < RB_T / 64; ++w) if (w < wave) woff += s_w[w];
    if (ok) {
        const int64_t pos = (int64_t)k * cap + block_base[(int64_t)k * nblk + blockIdx.x] + woff + rank;
        list_row[pos] = (int)r;
        list_val[pos] = v;
    }
}

// tile_starts[k][t] = lower_bound(rows[k][0..counts[k]), t*T)
__global__ __launch_bounds__(256) void tile_starts_k(const int32_t* __restrict__ rows, const int32_t* __restrict__ counts, int64_t cap,
                                                     int T, int64_t n_tiles, int32_t* out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;
    if (t > n_tiles) return;
    const int32_t* a = rows + (int64_t)k * cap;
    const int64_t target = t * T;
    int lo = 0, hi = counts[k];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)a[mid] < target) lo = mid + 1; else hi = mid;
    }
    out[(int64_t)k * (n_tiles + 1) + t] = lo;
}

static int compact(const int32_t* val, int64_t n, int K, int64_t cap, int32_t* list_row, int32_t* list_val,
                   int32_t* counts, int32_t* block_cnt, int32_t* block_base, hipStream_t s) {
    const int nblk = (int)ceil_div(n, RB_T);
    hipLaunchKernelGGL(rb_count_k, dim3(nblk, K), dim3(RB_T), 0, s, val, n, nblk, block_cnt);
    hipLaunchKernelGGL(rb_scan_k, dim3(K), dim3(RB_T), 0, s, (const int32_t*)block_cnt, nblk, block_base, counts);
    hipLaunchKernelGGL(rb_write_k, dim3(nblk, K), dim3(RB_T), 0, s, val, n, nblk, (const int32_t*)block_base, cap, list_row, list_val);
    return check_launch("rulebook compact");
}

static int64_t rb_ws_bytes(int64_t n, int K) {
    const int64_t nblk = ceil_div(n, RB_T);
    return (K * n + 2 * K * nblk + 64) * 4;
}

}  // namespace u3d

using namespace u3d;

extern "C" {

int64_t u3d_subm_rulebook_ws_bytes(int64_t n) { return (2 * 27 * ceil_div(n, RB_T) + 64) * 4; }      // block counts + bases only
int64_t u3d_down_rulebook_ws_bytes(int64_t n) { return rb_ws_bytes(n, 8); }

int u3d_subm_rulebook(const int32_t* coords, int64_t n, const uint64_t* bitmap, const int32_t* word_rank, int64_t hash_slots, int B,
                      int X, int Y, int Z, int32_t* pair_in, int32_t* pair_out, int32_t* counts, void* ws,
                      u3d_stream_t stream) {
    if (!coords || !bitmap || !word_rank || !pair_in || !pair_out || !counts || !ws || n <= 0) return U3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_RULEBOOK, s, 0.0);
    const Index ix = make_index(bitmap, word_rank, B, X, Y, Z, hash_slots);
    const int nblk = (int)ceil_div(n, RB_T);
    int32_t* bc = (int32_t*)ws;
    int32_t* bb = bc + (int64_t)27 * nblk;
    // rows = output voxel, value = input (neighbour) row; count -> per-offset scan of the block counts -> stable write
    hipLaunchKernelGGL(subm_count_k, dim3(nblk, 27), dim3(RB_T), 0, s, coords, n, ix, nblk, bc);
    hipLaunchKernelGGL(rb_scan_k, dim3(27), dim3(RB_T), 0, s, (const int32_t*)bc, nblk, bb, counts);
    hipLaunchKernelGGL(subm_write_k, dim3(nblk, 27), dim3(RB_T), 0, s, coords, n, ix, nblk, (const int32_t*)bb, n, pair_out, pair_in);
    return check_launch("subm rulebook");
}

int u3d_index_mark(const int32_t* coords, int64_t n, int shift, int X2, int Y2, int Z2, uint64_t* bitmap2, u3d_stream_t stream) {
    if (!coords || !bitmap2 || n <= 0 || X2 <= 0 || Y2 <= 0 || Z2 <= 0 || shift < 0 || shift > 8) return U3D_EINVAL;
    hipLaunchKernelGGL(index_mark_k, dim3((unsigned)ceil_div(n, RB_T)), dim3(RB_T), 0, (hipStream_t)stream, coords, n, shift, X2,
                       Y2, Z2, (Z2 + 63) / 64, (unsigned long long*)bitmap2);
    return check_launch("index_mark");
}

int u3d_down_rulebook(const int32_t* coords, int64_t n, const uint64_t* bitmap2, const int32_t* word_rank2, int64_t hash_slots, int B,
                      int X2, int Y2, int Z2, int32_t* pair_in, int32_t* pair_out, int32_t* counts, void* ws,
                      u3d_stream_t stream) {
    if (!coords || !bitmap2 || !word_rank2 || !pair_in || !pair_out || !counts || !ws || n <= 0) return U3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_RULEBOOK, s, 0.0);
    const Index ix = make_index(bitmap2, word_rank2, B, X2, Y2, Z2, hash_slots);
    const int nblk = (int)ceil_div(n, RB_T);
    int32_t* val = (int32_t*)ws;
    int32_t* bc = val + 8 * n;
    int32_t* bb = bc + (int64_t)8 * nblk;
    hipLaunchKernelGGL(down_nbr_k, dim3(nblk), dim3(RB_T), 0, s, coords, n, ix, val);
    // rows = input (child) voxel, value = output (parent) row
    return compact(val, n, 8, n, pair_in, pair_out, counts, bc, bb, s);
}

int u3d_tile_starts(const int32_t* rows, const int32_t* counts, int K, int64_t cap, int tile_rows, int64_t n_tiles,
                    int32_t* tile_starts, u3d_stream_t stream) {
    if (!rows || !counts || !tile_starts || K <= 0 || tile_rows <= 0 || n_tiles <= 0) return U3D_EINVAL;
    hipLaunchKernelGGL(tile_starts_k, dim3((unsigned)ceil_div(n_tiles + 1, 256), K), dim3(256), 0, (hipStream_t)stream, rows,
                       counts, cap, tile_rows, n_tiles, tile_starts);
    return check_launch("tile_starts");
}

}  // extern "C"
