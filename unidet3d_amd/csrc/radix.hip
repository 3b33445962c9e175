// Stable LSD radix sort of 64-bit keys (+ the permutation) for gfx950 -- hand-written, no library, no global atomics.
// Used by (a) the hashed voxel index (csrc/hashidx.hip: cell ids -> sorted unique cells -> canonical rows; the reference's hash-built
// rulebooks, unidet3d/unidet3d.py:158-174 / unidet3d/spconv_unet.py:43-56,148-154 through spconv / MinkowskiEngine) and (b) the CSR of
// points per superpoint (csrc/pool.hip: torch_scatter.scatter_mean at unidet3d/unidet3d.py:130, :332-333): sorting the point ids by
// segment id with a STABLE sort makes every segment's list ascending, so the fp32 sums over it have one order, run after run.
//
// One pass per 8-bit digit, three steps per pass:
//   rs_hist_k     a workgroup counts the digits of its 1024-key tile in LDS (integer LDS atomics: counts, order-free) and writes
//                 them DIGIT-major: hist[digit][tile];
//   exclusive scan of hist (misc.hip: the same scan the occupancy index uses): base[digit][tile] = first output slot of that
//                 tile's keys with that digit;
//   rs_scatter_k  the workgroup ranks its keys inside (tile, digit) in index order -- wave-level match by eight ballots (lanes
//                 with the same digit), a 16-row x 256-digit count table in LDS for the (item, wave) sub-tiles, one thread per digit
//                 turns it into prefixes -- and writes key (and value) to base + rank.  Stable by construction.
// Keys at or above `clamp` are written as `clamp` (the hashed index uses it to park its "no cell" sentinel one past the largest id
// so that the sort needs only the bits of the grid, not 63).
#include "u3d_common.h"

namespace u3d {

constexpr int RS_T = 256, RS_I = 4, RS_TILE = RS_T * RS_I;

__global__ __launch_bounds__(RS_T) void rs_hist_k(const uint64_t* __restrict__ kin, int64_t n, int shift, uint64_t clamp, int nblk,
                                                  int32_t* __restrict__ hist) {
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int j = 0; j < RS_I; ++j) {
        const int64_t i = t0 + j * RS_T + threadIdx.x;
        if (i < n) {
            uint64_t k = kin[i];
            k = k < clamp ? k : clamp;
            atomicAdd(&h[(int)((k >> shift) & 255)], 1);
        }
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// vin == nullptr: the value of key i is i (first pass of a sort that returns the permutation)
__global__ __launch_bounds__(RS_T) void rs_scatter_k(const uint64_t* __restrict__ kin, const int32_t* __restrict__ vin, int64_t n, int shift,
                                                     uint64_t clamp, int nblk, const int32_t* __restrict__ base, uint64_t* __restrict__ kout,
                                                     int32_t* __restrict__ vout) {
    __shared__ int cnt[RS_I * 4][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int r = 0; r < RS_I * 4; ++r) cnt[r][tid] = 0;
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * RS_TILE;
    uint64_t key[RS_I];
    int dig[RS_I], rk[RS_I];
    bool ok[RS_I];
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
    for (int j = 0; j < RS_I; ++j) {
        const int64_t i = t0 + j * RS_T + tid;
        ok[j] = i < n;
        uint64_t k = ok[j] ? kin[i] : 0ull;
        k = k < clamp ? k : clamp;
        key[j] = k;
        const int d = (int)((k >> shift) & 255);
        dig[j] = d;
        uint64_t peers = __ballot(ok[j]);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const uint64_t bal = __ballot((d >> b) & 1);
            peers &= ((d >> b) & 1) ? bal : ~bal;
        }
        rk[j] = __popcll(peers & below);
        if (ok[j] && rk[j] == 0) cnt[j * 4 + wave][d] = __popcll(peers);       // the lowest lane of a digit group writes its size
    }
    __syncthreads();
    {                                                        // thread = digit: counts -> output slots, sub-tiles in index order
        int run = base[(int64_t)tid * nblk + blockIdx.x];
#pragma unroll
        for (int r = 0; r < RS_I * 4; ++r) {
            const int c = cnt[r][tid];
            cnt[r][tid] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_I; ++j) {
        if (!ok[j]) continue;
        const int pos = cnt[j * 4 + wave][dig[j]] + rk[j];
        kout[pos] = key[j];
        if (vout) vout[pos] = vin ? vin[t0 + j * RS_T + tid] : (int32_t)(t0 + j * RS_T + tid);
    }
}

// ---- unique of a sorted array: flags -> scan -> compaction ----
__global__ __launch_bounds__(256) void uniq_flag_k(const uint64_t* __restrict__ sorted, int64_t n, int32_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || sorted[i] != sorted[i - 1]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void uniq_write_k(const uint64_t* __restrict__ sorted, const int32_t* __restrict__ pos /*[n+1] exclusive*/, int64_t n,
                                                    uint64_t drop, uint64_t* __restrict__ out, int32_t* __restrict__ n_unique) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = sorted[i];
    if (pos[i + 1] != pos[i]) out[pos[i]] = k;
    if (i == n - 1) *n_unique = pos[n] - (k == drop ? 1 : 0);          // the last (largest) key is the parked sentinel: not counted
}

static inline int64_t al256(int64_t b) { return (b + 255) & ~(int64_t)255; }

int64_t radix_ws_bytes(int64_t n, bool with_values) {
    const int64_t nblk = ceil_div(n, RS_TILE);
    return al256(n * 8) + (with_values ? al256(n * 4) : 0) + 2 * al256((256 * nblk + 1) * 4) + al256(scan_ws_bytes(256 * nblk)) + 256;
}

// keys_in [n] -> keys_out [n] ascending (stable); vals_out (nullable) [n] = index of each sorted key in keys_in.  `bits`: number of
// significant low key bits AFTER clamping (passes = ceil(bits / 8)).  keys_in is not modified; keys_out / vals_out may not alias it.
int radix_sort_u64(const uint64_t* keys_in, int64_t n, int bits, uint64_t clamp, uint64_t* keys_out, int32_t* vals_out, void* ws, hipStream_t s) {
    if (n <= 0 || n >= 0x7fffffffLL || bits < 1 || bits > 64) return U3D_EINVAL;
    const int passes = (bits + 7) / 8;
    const int nblk = (int)ceil_div(n, RS_TILE);
    char* w = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    uint64_t* ktmp = (uint64_t*)w; w += al256(n * 8);
    int32_t* vtmp = nullptr;
    if (vals_out) { vtmp = (int32_t*)w; w += al256(n * 4); }
    int32_t* hist = (int32_t*)w; w += al256((256 * (int64_t)nblk + 1) * 4);
    int32_t* base = (int32_t*)w; w += al256((256 * (int64_t)nblk + 1) * 4);
    void* sws = w;
    // ping-pong so that the LAST pass writes keys_out / vals_out
    const uint64_t* kin = keys_in;
    const int32_t* vin = nullptr;
    for (int p = 0; p < passes; ++p) {
        const bool to_out = ((passes - 1 - p) & 1) == 0;
        uint64_t* ko = to_out ? keys_out : ktmp;
        int32_t* vo = vals_out ? (to_out ? vals_out : vtmp) : nullptr;
        hipLaunchKernelGGL(rs_hist_k, dim3((unsigned)nblk), dim3(RS_T), 0, s, kin, n, p * 8, clamp, nblk, hist);
        int rc = exclusive_scan_i32(hist, 256 * (int64_t)nblk, base, sws, s);
        if (rc) return rc;
        hipLaunchKernelGGL(rs_scatter_k, dim3((unsigned)nblk), dim3(RS_T), 0, s, kin, vin, n, p * 8, clamp, nblk, (const int32_t*)base, ko, vo);
        kin = ko;
        vin = vo;
    }
    return check_launch("radix_sort");
}

int64_t unique_ws_bytes(int64_t n) { return 2 * al256((n + 1) * 4) + al256(scan_ws_bytes(n)) + 256; }

// sorted [n] ascending -> out: the distinct keys in order; *n_unique = their number, not counting a trailing key equal to `drop`
int unique_sorted_u64(const uint64_t* sorted, int64_t n, uint64_t drop, uint64_t* out, int32_t* n_unique, void* ws, hipStream_t s) {
    if (n <= 0 || n >= 0x7fffffffLL) return U3D_EINVAL;
    char* w = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    int32_t* flag = (int32_t*)w; w += al256((n + 1) * 4);
    int32_t* pos = (int32_t*)w; w += al256((n + 1) * 4);
    const unsigned g = (unsigned)ceil_div(n, 256);
    hipLaunchKernelGGL(uniq_flag_k, dim3(g), dim3(256), 0, s, sorted, n, flag);
    int rc = exclusive_scan_i32(flag, n, pos, w, s);
    if (rc) return rc;
    hipLaunchKernelGGL(uniq_write_k, dim3(g), dim3(256), 0, s, sorted, (const int32_t*)pos, n, drop, out, n_unique);
    return check_launch("unique_sorted");
}

}  // namespace u3d

using namespace u3d;

extern "C" {

int64_t u3d_sort_ws_bytes(int64_t n, int with_values) {
    if (n <= 0 || n >= 0x7fffffffLL) return 0;
    return radix_ws_bytes(n, with_values != 0);
}

int u3d_sort_u64(const uint64_t* keys_in, int64_t n, int key_bits, uint64_t* keys_out, int32_t* perm_out, void* ws, u3d_stream_t stream) {
    if (!keys_in || !keys_out || !ws || n <= 0 || n >= 0x7fffffffLL || key_bits < 1 || key_bits > 64) return U3D_EINVAL;
    if ((const void*)keys_in == (const void*)keys_out) { set_error("sort_u64: keys_out must not alias keys_in"); return U3D_EINVAL; }
    ProfScope prof(U3D_K_RULEBOOK, (hipStream_t)stream, 0.0);
    return radix_sort_u64(keys_in, n, key_bits, ~0ull, keys_out, perm_out, ws, (hipStream_t)stream);
}

}  // extern "C"
