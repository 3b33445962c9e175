// R1 voxelisation on gfx950: occupancy bitmap + popcount rank instead of a hash table.
// Replaces ME.utils.batch_sparse_collate / ME.TensorField.sparse() / inverse_mapping
// (reference call site unidet3d/unidet3d.py:158-174).  HBM-bound integer work:
// every pass streams the point array once with coalesced reads; the bitmap (a few MB at
// ScanNet extents) lives in L2 / Infinity Cache.
#include "u3d_common.h"

namespace u3d {

__device__ __forceinline__ int f2ord(float f) {
    int b = __float_as_int(f);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float ord2f(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

__device__ __forceinline__ int cell_of(float v, float mn, float vs, float inv_vs, int mode) {
    const float d = v - mn;
    const float c = mode == 0 ? d / vs : d * inv_vs;
    return (int)floorf(c);
}

struct StatsWs {          // per scene
    int mn[3], mx[3];
    double sum[3];
    int pad[2];
};

__global__ void stats_init_k(StatsWs* ws, int B, int32_t* grid_max) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        for (int a = 0; a < 3; ++a) {
            ws[b].mn[a] = 0x7fffffff;
            ws[b].mx[a] = (int)0x80000000;
            ws[b].sum[a] = 0.0;
        }
    }
    if (b < 3) grid_max[b] = 0;
}

// min / max: integer atomics on order-preserving keys (order-free).  Coordinate sums: every (scene, block) writes ONE fp64 partial
// (its four waves added in a fixed order), stats_fin_k adds a scene's partials in block order -- no floating-point atomics: the scene
// mean, and with it every voxel feature, has the same bits run after run.
constexpr int STATS_MAX_BLK = 256;
__global__ __launch_bounds__(256) void scene_stats_k(const float* __restrict__ points, const float* __restrict__ csrc,
                                                     const int64_t* __restrict__ offs, StatsWs* ws, double* __restrict__ part) {
    const int b = blockIdx.y;
    const int64_t lo = offs[b], hi = offs[b + 1];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    double sm[3] = {0, 0, 0};
    for (int64_t p = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < hi; p += (int64_t)gridDim.x * blockDim.x) {
        const float* q = points + p * 6;
        const float x = q[0], y = q[1], z = q[2];
        sm[0] += x; sm[1] += y; sm[2] += z;
        float cx = x, cy = y, cz = z;
        if (csrc) { cx = csrc[p * 3]; cy = csrc[p * 3 + 1]; cz = csrc[p * 3 + 2]; }
        mn[0] = fminf(mn[0], cx); mn[1] = fminf(mn[1], cy); mn[2] = fminf(mn[2], cz);
        mx[0] = fmaxf(mx[0], cx); mx[1] = fmaxf(mx[1], cy); mx[2] = fmaxf(mx[2], cz);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], d, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d, 64));
            sm[a] += __shfl_xor(sm[a], d, 64);
        }
    }
    __shared__ double wsum[4][3];
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) wsum[threadIdx.x >> 6][a] = sm[a];
        if (lo < hi) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                atomicMin(&ws[b].mn[a], f2ord(mn[a]));
                atomicMax(&ws[b].mx[a], f2ord(mx[a]));
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 3)
        part[((int64_t)b * STATS_MAX_BLK + blockIdx.x) * 3 + threadIdx.x] =
            ((wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + wsum[2][threadIdx.x]) + wsum[3][threadIdx.x];
}

__global__ void stats_fin_k(const StatsWs* ws, const double* __restrict__ part, int nblk, const int64_t* offs, int B, float vs, float inv_vs,
                            int mode, float* stats, int32_t* grid_max) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double n = (double)(offs[b + 1] - offs[b]);
    for (int a = 0; a < 3; ++a) {
        float mn = ord2f(ws[b].mn[a]), mx = ord2f(ws[b].mx[a]);
        stats[b * 12 + a] = mn;
        stats[b * 12 + 3 + a] = mx;
        double sum = 0.0;
        for (int k = 0; k < nblk; ++k) sum += part[((int64_t)b * STATS_MAX_BLK + k) * 3 + a];
        stats[b * 12 + 6 + a] = n > 0 ? (float)(sum / n) : 0.f;
        stats[b * 12 + 9 + a] = 0.f;
        if (n > 0) atomicMax(&grid_max[a], cell_of(mx, mn, vs, inv_vs, mode));
    }
}

__global__ __launch_bounds__(256) void vox_mark_k(const float* __restrict__ points, const float* __restrict__ csrc,
                                                  const int64_t* __restrict__ offs, const float* __restrict__ stats,
                                                  float vs, float inv_vs, int mode, int X, int Y, int Z, int Zw,
                                                  unsigned long long* bitmap, int64_t* pt_cell) {
    const int b = blockIdx.y;
    const int64_t p = offs[b] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= offs[b + 1]) return;
    float c[3];
    if (csrc) { c[0] = csrc[p * 3]; c[1] = csrc[p * 3 + 1]; c[2] = csrc[p * 3 + 2]; }
    else { c[0] = points[p * 6]; c[1] = points[p * 6 + 1]; c[2] = points[p * 6 + 2]; }
    int x = cell_of(c[0], stats[b * 12 + 0], vs, inv_vs, mode);
    int y = cell_of(c[1], stats[b * 12 + 1], vs, inv_vs, mode);
    int z = cell_of(c[2], stats[b * 12 + 2], vs, inv_vs, mode);
    x = min(max(x, 0), X - 1); y = min(max(y, 0), Y - 1); z = min(max(z, 0), Z - 1);
    const int64_t w = ((int64_t)(b * X + x) * Y + y) * Zw + (z >> 6);
    if (bitmap) atomicOr(&bitmap[w], 1ull << (z & 63));       // (hashed index: only the cell ids are wanted)
    pt_cell[p] = w * 64 + (z & 63);
}

__global__ __launch_bounds__(256) void index_coords_k(const uint64_t* __restrict__ bitmap, const int32_t* __restrict__ rank,
                                                      int64_t n_words, int X, int Y, int Zw, int32_t* coords) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint64_t word = bitmap[w];
    if (!word) return;
    int row = rank[w];
    const int zw = (int)(w % Zw);
    int64_t t = w / Zw;
    const int y = (int)(t % Y); t /= Y;
    const int x = (int)(t % X);
    const int b = (int)(t / X);
    while (word) {
        const int bit = __ffsll((long long)word) - 1;
        word &= word - 1;
        int4 c = make_int4(b, x, y, zw * 64 + bit);
        *reinterpret_cast<int4*>(coords + (int64_t)row * 4) = c;
        ++row;
    }
}

__global__ __launch_bounds__(256) void vox_inverse_k(const int64_t* __restrict__ pt_cell, Index ix, int64_t n_pts, int64_t* inverse,
                                                     int32_t* cnt) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pts) return;
    const int row = index_row_of_cell(ix, pt_cell[p]);
    inverse[p] = row;
    atomicAdd(&cnt[row], 1);
}

__global__ __launch_bounds__(256) void vox_fill_k(const int64_t* __restrict__ inverse, const int32_t* __restrict__ offsets,
                                                  int64_t n_pts, int32_t* cursor, int32_t* list) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pts) return;
    const int row = (int)inverse[p];
    const int pos = offsets[row] + atomicAdd(&cursor[row], 1);
    list[pos] = (int)p;
}

// one thread per voxel: order its (short) point list by point id -> deterministic, then mean in fp64
__global__ __launch_bounds__(256) void vox_feats_k(const float* __restrict__ points, const int64_t* __restrict__ offs, int B,
                                                   const float* __restrict__ stats, const int32_t* __restrict__ vox_offsets,
                                                   int32_t* vox_points, int64_t n_vox, float* feats, int ld) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vox) return;
    const int lo = vox_offsets[v], hi = vox_offsets[v + 1];
    for (int i = lo + 1; i < hi; ++i) {          // insertion sort, lists are a handful of points
        int key = vox_points[i], j = i - 1;
        while (j >= lo && vox_points[j] > key) { vox_points[j + 1] = vox_points[j]; --j; }
        vox_points[j + 1] = key;
    }
    if (lo >= hi) return;
    const int64_t p0 = vox_points[lo];
    int b = 0;
    while (b + 1 < B && offs[b + 1] <= p0) ++b;   // all points of a voxel belong to one scene
    const float mx = stats[b * 12 + 6], my = stats[b * 12 + 7], mz = stats[b * 12 + 8];
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int i = lo; i < hi; ++i) {
        const float* q = points + (int64_t)vox_points[i] * 6;
        s[0] += q[3]; s[1] += q[4]; s[2] += q[5];
        s[3] += q[0] - mx; s[4] += q[1] - my; s[5] += q[2] - mz;
    }
    const double inv = 1.0 / (double)(hi - lo);
#pragma unroll
    for (int c = 0; c < 6; ++c) feats[v * ld + c] = (float)(s[c] * inv);
}

}  // namespace u3d

using namespace u3d;

extern "C" {

int64_t u3d_vox_scene_stats_ws_bytes(int B) { return (int64_t)B * sizeof(StatsWs) + (int64_t)B * STATS_MAX_BLK * 3 * sizeof(double) + 128; }

int u3d_vox_scene_stats(const float* points, const float* coord_src, const int64_t* pt_offsets, int B,
                        int64_t max_pts, float voxel_size, int div_mode, float* stats, int32_t* grid_max,
                        void* ws, u3d_stream_t stream) {
    if (!points || !pt_offsets || B <= 0 || !stats || !grid_max || !ws || voxel_size <= 0.f) return U3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_VOXELIZE, s, 0.0);
    StatsWs* w = (StatsWs*)ws;
    hipLaunchKernelGGL(stats_init_k, dim3((B + 63) / 64 + 1), dim3(64), 0, s, w, B, grid_max);
    int nblk = (int)ceil_div(max_pts, 256 * 8);
    nblk = nblk < 1 ? 1 : (nblk > STATS_MAX_BLK ? STATS_MAX_BLK : nblk);
    double* part = (double*)(((uintptr_t)(w + B) + 63) & ~(uintptr_t)63);
    hipLaunchKernelGGL(scene_stats_k, dim3(nblk, B), dim3(256), 0, s, points, coord_src, pt_offsets, w, part);
    hipLaunchKernelGGL(stats_fin_k, dim3((B + 63) / 64), dim3(64), 0, s, (const StatsWs*)w, (const double*)part, nblk, pt_offsets, B,
                       voxel_size, 1.0f / voxel_size, div_mode, stats, grid_max);
    return check_launch("vox_scene_stats");
}

// The index is a direct-address table: its size follows the EXTENT of the grid, not the number of occupied voxels (12 bytes per
// 64 cells along z).  Indoor scans at 2 cm stay in the tens of MB (8 scenes x 512 x 512 x 256 cells = 100 MB); a grid whose
// table would pass U3D_INDEX_MAX_WORDS (2^31 words = 24 GiB of bitmap + rank) is refused instead of allocated -- outdoor-scale
// extents need a sparse (hashed or two-level) index, which this library does not have.
int64_t u3d_index_words(int B, int X, int Y, int Z) {
    if (B <= 0 || X <= 0 || Y <= 0 || Z <= 0) return U3D_EINVAL;
    const int64_t zw = (Z + 63) / 64;
    const long double words = (long double)B * X * Y * zw;
    if (words > (long double)U3D_INDEX_MAX_WORDS) {
        set_error("occupancy index: grid %d x %d x %d x %d needs %.3Lg words (limit %lld): extent too large for the direct-address index",
                  B, X, Y, Z, words, (long long)U3D_INDEX_MAX_WORDS);
        return U3D_EUNSUPPORTED;
    }
    return (int64_t)B * X * Y * zw;
}

int u3d_vox_mark(const float* points, const float* coord_src, const int64_t* pt_offsets, int B, int64_t max_pts,
                 const float* stats, float voxel_size, int div_mode, int X, int Y, int Z, uint64_t* bitmap,
                 int64_t* pt_cell, u3d_stream_t stream) {
    if (!points || !pt_offsets || !stats || !pt_cell || B <= 0 || X <= 0 || Y <= 0 || Z <= 0) return U3D_EINVAL;      // bitmap may be NULL
    if ((long double)B * X * Y * ((Z + 63) / 64) * 64 >= 4.0e18L) { set_error("vox_mark: grid %d x %d x %d x %d exceeds 62-bit cell ids", B, X, Y, Z); return U3D_EUNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_VOXELIZE, s, 0.0);
    if (max_pts <= 0) return U3D_OK;
    hipLaunchKernelGGL(vox_mark_k, dim3((unsigned)ceil_div(max_pts, 256), B), dim3(256), 0, s, points, coord_src,
                       pt_offsets, stats, voxel_size, 1.0f / voxel_size, div_mode, X, Y, Z, (Z + 63) / 64,
                       (unsigned long long*)bitmap, pt_cell);
    return check_launch("vox_mark");
}

int u3d_index_coords(const uint64_t* bitmap, const int32_t* word_rank, int B, int X, int Y, int Z, int32_t* coords,
                     u3d_stream_t stream) {
    if (!bitmap || !word_rank || !coords) return U3D_EINVAL;
    const int Zw = (Z + 63) / 64;
    const int64_t nw = (int64_t)B * X * Y * Zw;
    hipLaunchKernelGGL(index_coords_k, dim3((unsigned)ceil_div(nw, 256)), dim3(256), 0, (hipStream_t)stream, bitmap,
                       word_rank, nw, X, Y, Zw, coords);
    return check_launch("index_coords");
}

int64_t u3d_vox_finalize_ws_bytes(int64_t n_pts, int64_t n_vox) {
    return (n_vox + 1) * 4 * 2 + 256 + scan_ws_bytes(n_vox);
}

int u3d_vox_finalize(const float* points, const int64_t* pt_offsets, int B, int64_t n_pts, const float* stats,
                     const int64_t* pt_cell, const uint64_t* bitmap, const int32_t* word_rank, int64_t hash_slots, int64_t n_vox,
                     int64_t* inverse, int32_t* vox_offsets, int32_t* vox_points, float* feats, int feat_ld, void* ws,
                     u3d_stream_t stream) {
    if (!points || !pt_offsets || !stats || !pt_cell || !bitmap || !word_rank || !inverse || !vox_offsets ||
        !vox_points || !feats || !ws || feat_ld < 6)
        return U3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_VOXELIZE, s, 0.0);
    if (n_pts <= 0 || n_vox <= 0) return U3D_OK;
    int32_t* cnt = (int32_t*)ws;
    int32_t* cursor = cnt + (n_vox + 1);
    void* sws = (void*)(((uintptr_t)(cursor + n_vox + 1) + 63) & ~(uintptr_t)63);
    hipMemsetAsync(cnt, 0, (size_t)(2 * (n_vox + 1)) * 4, s);
    const unsigned gp = (unsigned)ceil_div(n_pts, 256);
    // (only the table / bitmap pointers of the index are used here: cell ids carry the geometry)
    hipLaunchKernelGGL(vox_inverse_k, dim3(gp), dim3(256), 0, s, pt_cell, make_index(bitmap, word_rank, B, 1, 1, 1, hash_slots), n_pts, inverse, cnt);
    int rc = exclusive_scan_i32(cnt, n_vox, vox_offsets, sws, s);
    if (rc) return rc;
    hipLaunchKernelGGL(vox_fill_k, dim3(gp), dim3(256), 0, s, (const int64_t*)inverse, (const int32_t*)vox_offsets,
                       n_pts, cursor, vox_points);
    hipLaunchKernelGGL(vox_feats_k, dim3((unsigned)ceil_div(n_vox, 256)), dim3(256), 0, s, points, pt_offsets, B, stats,
                       (const int32_t*)vox_offsets, vox_points, n_vox, feats, feat_ld);
    return check_launch("vox_finalize");
}

}  // extern "C"
