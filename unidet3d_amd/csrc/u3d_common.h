// Internal helpers shared by the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "u3d.h"

namespace u3d {

void set_error(const char* fmt, ...);

// ---- per-class kernel timing with HIP events on the launch stream --------------------
struct ProfScope {
    int cls;
    hipStream_t stream;
    bool on;
    hipEvent_t e0, e1;
    double work;
    ProfScope(int cls, hipStream_t s, double work);
    ~ProfScope();
};

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return U3D_ELAUNCH;
    }
    return U3D_OK;
}

__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// "this kernel's dynamic-LDS attribute is set on device d": one flag per device ordinal, read and written from any host thread
// (autograd runs a worker per device); ordinals past the table always set the attribute again (`dev & 63` aliased them)
struct DeviceOnce {
    std::atomic<bool> flag[64];
    bool needed(int dev) const { return (unsigned)dev >= 64u || !flag[dev].load(std::memory_order_acquire); }
    void done(int dev) { if ((unsigned)dev < 64u) flag[dev].store(true, std::memory_order_release); }
};

// XCD-aware work-id remap (MI355X: 8 XCDs, workgroup b is dispatched to XCD b % 8, each XCD has its own
// 4 MB L2): XCD x gets the CONTIGUOUS chunk of work ids so that neighbouring tiles, which gather the same
// feature rows, share an L2.  Bijective for any n (speed only -- correctness never depends on placement).
__device__ __forceinline__ int64_t xcd_swizzle(int64_t b, int64_t n) {
    const int64_t q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// ---- raw buffer loads (32-bit offsets from a scalar base: one VALU instruction or none per address) ------------------
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ __forceinline__ f32x4 bload128(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ int bload32(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return (int)__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
}
// raw dword buffer of `bytes` bytes (default: a 2 GiB window): loads past the end return 0, stores past the end are dropped
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, int64_t bytes = 0x7fffffff) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes < 0x7fffffff ? bytes : 0x7fffffff), 0x00020000);
}
#endif

// ---- fp32 products on the bf16 matrix pipe ("bf16x3") -------------------------------------------------------------------
// The fp32 MFMAs of gfx950 run at the fp32 VALU rate (157 TFLOP/s, 32 MAC / cycle / SIMD); the bf16 ones at 16x that.  An fp32
// value splits EXACTLY into three bf16 pieces x = h + m + l: h = x rounded to 8 significant bits, m = x - h truncated to 8 bits,
// l = x - h - m; both subtractions are exact and l has at most 8 significant bits, so it is a bf16 value.  A product becomes
//     x y = hx hy + (hx my + mx hy) + (hx ly + mx my + lx hy) + [mx ly + lx my + lx ly, dropped: ~2^-26 |x y|, either sign]:
// six bf16 MFMAs (products of 8-bit significands are exact in the fp32 accumulator) instead of sixteen MFMA-equivalents of fp32
// issue time, with the error of an fp32 FMA chain (tests/test_gpu_kernels.py / test_gpu_model.py measure both modes against
// fp64, per kernel and end to end).  U3D_FP32_MATH=mfma selects the native fp32 MFMA kernels instead.
extern int g_fp32_math;
extern int g_conv_kernel;      // 1: workgroup-tile sparse convolution (spconv_wg.hip), 0: wave tiles (u3d_conv_kernel)
bool fp32_x3();
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
using u16x8 = __attribute__((ext_vector_type(8))) unsigned short;
using bf16x8_t = __attribute__((ext_vector_type(8))) __bf16;
// two fp32 values -> one dword per plane (low half: a, high half: b).  h is x ROUNDED to 8 significant bits (add half an ulp to
// the bit pattern, clear the low 16 bits: round to nearest, ties away), m is x - h TRUNCATED to 8 bits, l the exact rest:
// |x - h| <= 2^-8 |x| with either sign, |l| < 2^-7 |x - h|, so the dropped cross terms m.l + l.m are <= 2^-22 of a product in the
// worst case, ~2^-26 on average, and of either sign -- the level of an fp32 multiply-add's own rounding.  13 VALU instructions
// per pair.  Measured alternatives (tools/split_rate.hip, SIMD cycles per pair incl. the loop body: 50 / 65 / 65):
//   * truncating h as well (11 instructions): remainders twice as large and all of the product's sign -- dropped terms up to
//     2^-20, biased towards zero; 2.5x larger backbone-gradient errors end to end (tests/test_gpu_full_size.py cfg4 over 1e-3);
//   * v_cvt_pk_bf16_f32 for both levels (round to nearest even, 11 instructions but slower ones), or add-half on both levels
//     (15): attention +22 %, GEMM +10 % kernel time for errors this variant already brings to the fp32 level.  Round 4 retried
//     the conversion form at its minimum -- v_cvt_pk_bf16_f32 through asm (no re-conversion per use) and ONE v_pk_add_f32 per
//     level: 9 instructions per pair, 14 % faster than this one in an isolated dependent loop (tools/split_bench.hip), but
//     attention +5 %, GEMM +1.5 %, convolution +-0 in the bench step: the conversion is not a full-rate instruction.
__device__ __forceinline__ void split3_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned ah = (__builtin_bit_cast(unsigned, a) + 0x8000u) & 0xffff0000u, bh = (__builtin_bit_cast(unsigned, b) + 0x8000u) & 0xffff0000u;
    const float a1 = a - __builtin_bit_cast(float, ah), b1 = b - __builtin_bit_cast(float, bh);
    const unsigned a1b = __builtin_bit_cast(unsigned, a1), b1b = __builtin_bit_cast(unsigned, b1);
    const float a2 = a1 - __builtin_bit_cast(float, a1b & 0xffff0000u), b2 = b1 - __builtin_bit_cast(float, b1b & 0xffff0000u);
    h = __builtin_amdgcn_perm(bh, ah, 0x07060302u);
    m = __builtin_amdgcn_perm(b1b, a1b, 0x07060302u);
    l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b2), __builtin_bit_cast(unsigned, a2), 0x07060302u);
}
// eight fp32 values -> the three bf16x8 planes
__device__ __forceinline__ void split3_x8(const f32x4& lo, const f32x4& hi, bf16x8_t (&out)[3]) {
#ifdef U3D_SPLIT_ABL       // timing ablation (tools/build_variant.sh <file> -DU3D_SPLIT_ABL): operands without the split arithmetic, WRONG results
    out[0] = __builtin_bit_cast(bf16x8_t, lo); out[1] = __builtin_bit_cast(bf16x8_t, hi); out[2] = out[0];
    return;
#endif
    unsigned w[3][4];
    split3_pair(lo[0], lo[1], w[0][0], w[1][0], w[2][0]);
    split3_pair(lo[2], lo[3], w[0][1], w[1][1], w[2][1]);
    split3_pair(hi[0], hi[1], w[0][2], w[1][2], w[2][2]);
    split3_pair(hi[2], hi[3], w[0][3], w[1][3], w[2][3]);
    u32x4 p[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) p[q] = u32x4{w[q][0], w[q][1], w[q][2], w[q][3]};
#pragma unroll
    for (int q = 0; q < 3; ++q) out[q] = __builtin_bit_cast(bf16x8_t, p[q]);
}
#endif

// exclusive scan of n int32 values produced by a functor; out has n+1 entries (out[n] = total)
int exclusive_scan_popc64(const uint64_t* words, int64_t n, int32_t* out, void* ws, hipStream_t s);
int exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, void* ws, hipStream_t s);
int64_t scan_ws_bytes(int64_t n);

// ---- csrc/radix.hip: stable LSD radix sort (8-bit digits) and unique of a sorted array; hand-written, deterministic ----
// keys at or above `clamp` sort as `clamp`; `bits` = significant low bits of the (clamped) keys; vals_out (nullable) = the permutation
int64_t radix_ws_bytes(int64_t n, bool with_values);
int radix_sort_u64(const uint64_t* keys_in, int64_t n, int bits, uint64_t clamp, uint64_t* keys_out, int32_t* vals_out, void* ws, hipStream_t s);
int64_t unique_ws_bytes(int64_t n);
int unique_sorted_u64(const uint64_t* sorted, int64_t n, uint64_t drop, uint64_t* out, int32_t* n_unique, void* ws, hipStream_t s);

// ---- occupancy index (bitmap + popcount rank) ------------------------------------------
// Key -> canonical row map of one level.  Two forms behind one lookup:
//  * direct-address (default): occupancy bitmap + popcount rank per 64 z-cells; size follows the grid EXTENT;
//  * hashed (hkeys != nullptr; csrc/hashidx.hip): open-addressing table cell id -> row built from the sorted unique cell ids;
//    size follows the number of occupied voxels -- for extents where the bitmap would be impractical (outdoor scenes).
// Cell id of (b, x, y, z) = ((b X + x) Y + y) Zw 64 + z in both forms (= bitmap word * 64 + bit): ascending id = canonical order.
struct Index {
    const uint64_t* bitmap;
    const int32_t* rank;
    int B, X, Y, Z, Zw;
    const uint64_t* hkeys = nullptr;      // hashed form: table of cell ids (U3D_HASH_EMPTY = free slot) ...
    const int32_t* hvals = nullptr;       // ... and their rows
    uint64_t hmask = 0;                   // slots - 1 (slots: power of two)
};

constexpr uint64_t U3D_HASH_EMPTY = ~0ull;
__device__ __forceinline__ uint64_t hash_mix(uint64_t k) {      // murmur3 finaliser
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}
__device__ __forceinline__ int hash_find(const Index& ix, uint64_t cell) {
    uint64_t slot = hash_mix(cell) & ix.hmask;
    while (true) {
        const uint64_t k = ix.hkeys[slot];
        if (k == cell) return ix.hvals[slot];
        if (k == U3D_HASH_EMPTY) return -1;
        slot = (slot + 1) & ix.hmask;
    }
}
static inline Index make_index(const uint64_t* bitmap_or_keys, const int32_t* rank_or_vals, int B, int X, int Y, int Z, int64_t hash_slots) {
    Index ix;
    ix.B = B; ix.X = X; ix.Y = Y; ix.Z = Z; ix.Zw = (Z + 63) / 64;
    if (hash_slots > 0) { ix.bitmap = nullptr; ix.rank = nullptr; ix.hkeys = bitmap_or_keys; ix.hvals = rank_or_vals; ix.hmask = (uint64_t)hash_slots - 1; }
    else { ix.bitmap = bitmap_or_keys; ix.rank = rank_or_vals; }
    return ix;
}

__device__ __forceinline__ int index_lookup(const Index& ix, int b, int x, int y, int z) {
    if ((unsigned)x >= (unsigned)ix.X || (unsigned)y >= (unsigned)ix.Y || (unsigned)z >= (unsigned)ix.Z) return -1;
    const int64_t w = ((int64_t)(b * ix.X + x) * ix.Y + y) * ix.Zw + (z >> 6);
    const int bit = z & 63;
    if (ix.hkeys) return hash_find(ix, (uint64_t)w * 64 + bit);
    const uint64_t word = ix.bitmap[w];
    if (!((word >> bit) & 1ull)) return -1;
    return ix.rank[w] + __popcll(word & ((1ull << bit) - 1ull));
}
// row of a cell id (bitmap word * 64 + bit) that is known to be occupied
__device__ __forceinline__ int index_row_of_cell(const Index& ix, int64_t cell) {
    if (ix.hkeys) return hash_find(ix, (uint64_t)cell);
    const int64_t w = cell >> 6;
    const int bit = (int)(cell & 63);
    return ix.rank[w] + __popcll(ix.bitmap[w] & ((1ull << bit) - 1ull));
}

}  // namespace u3d
