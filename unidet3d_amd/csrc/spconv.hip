// K4-K8 sparse convolutions on gfx950: one output-stationary gather-MFMA-scatter kernel for
// SubMConv3d / SparseConv3d / SparseInverseConv3d forward and input-gradient, one pair-stationary
// MFMA kernel for the weight gradient.  Replaces spconv's implicit-GEMM kernels behind
// unidet3d/spconv_unet.py:34-72,146-192 and unidet3d/unidet3d.py:96-103.
//
// Forward / dgrad (spconv_gmm_k), wave-independent design (no workgroup barriers at all):
//   * every wave64 owns R (32 or 64) consecutive DST rows x a 32-column slice of the output and,
//     optionally, one of G groups of kernel offsets; its fp32 accumulator tile [R+1][40] is private
//     LDS, so dst is written exactly once per (row, column) -- no global atomics, no per-pair
//     feature round trip through HBM (algorithmic traffic N*(Cs+Cd)*4 + pair indices + weights).
//   * the canonical rulebook is the working structure: for offset k the wave's pairs are the
//     contiguous range tile_starts[k][t] .. tile_starts[k][t+1] of the ascending scatter list;
//     offsets without a pair in the tile cost two lane reads, absent neighbours cost nothing.
//   * per offset the wave takes windows of 16 or 32 pairs: whole source rows are loaded with buffer
//     loads, transposed to MFMA fragments through a private LDS image, multiplied with the packed
//     weight fragments (u3d_weight_pack) by v_mfma_f32_16x16x4_f32 in the TRANSPOSED orientation
//     and accumulated straight through the MFMA C operand into the LDS tile (see GmmWave below).
//   * 4 independent waves per workgroup, ~12 KB LDS per wave; deep U-Net levels (a few thousand
//     rows) get their parallelism from column slices and offset groups (partials summed by a small
//     deterministic reduce kernel).
//   * fp32 in / fp32 accumulate MFMA (exact fp32, 157 TF peak) -- BASELINE config 2 is fp32.
#include <stdlib.h>

#include "u3d_common.h"
#include "spconv_gmm.h"

namespace u3d {

// Timing ablations (tools/build_variant.sh ... -DU3D_GMM_ABL=n; results are WRONG by construction, never shipped):
//   bit 0 (1) = no MFMAs (operands kept alive), bit 1 (2) = no accumulator read-modify-write in LDS, 4 = every gather hits rows 0..63,
//   bit 3 (8) = no LDS staging transposition, bit 4 (16) = no index shuffles, bit 5 (32) = weights loaded once
#ifndef U3D_GMM_ABL
#define U3D_GMM_ABL 0
#endif
#ifndef U3D_WG_ABL
#define U3D_WG_ABL 0
#endif
// float index of (row r, 16-byte quad c4) in the tile
__device__ __forceinline__ int gmm_acc_idx(int r, int c4) { return GMM_SWZ ? r * 32 + ((c4 ^ (r & 7)) << 2) : r * GMM_ALD + (c4 << 2); }

// ---- software-pipelined wave program ---------------------------------------------------------------------
// A wave's work is a sequence of ITEMS (offset k, window of W = 16*NCH pairs of k's range in this row tile);
// an item is computed in NJB UNITS of JB 16-channel groups of the source channels.  Per unit the wave holds
// the gathered rows and the packed weight fragments in registers; the loads of unit u+1 are issued BEFORE the
// MFMAs of unit u and the raw gather/scatter indices are loaded two items ahead.  Two register buffers
// alternate; the unit sequence is unrolled so buffer roles are static.  What shaped the rest (all measured on
// MI355X, tools/coissue.hip, tools/l1_bw.hip, tools/mfma_peak.hip):
//  * VALU instructions do NOT overlap with fp32 MFMAs of other waves on the same SIMD (MFMA loop 3.7 ms, fp32
//    VALU loop 2.0 ms, both together 5.3 ms; LDS traffic does overlap).  SIMD time = 32 cycles per MFMA + 4-16 per
//    VALU instruction, so the program is written for VALU count: buffer loads (one v_mad_u32_u24 per gathered
//    row instead of 64-bit address chains), scalar weight offsets, per-lane constants hoisted, and an epilogue
//    without VALU arithmetic (below).  ~20 VALU instructions per 32-pair item against 16-32 MFMAs.
//  * The MFMA computes the TRANSPOSED tile (A operand = weight fragment, B operand = gathered rows), so a lane
//    ends up with 4 consecutive output columns of ONE pair: the accumulator row is read with a single
//    ds_read_b128 straight into the MFMA's C operand and written back with one ds_write_b128 -- the MFMA does
//    the accumulation, there is no v_add, no zero fill, and one index shuffle per chunk instead of four.
//    (Inside one offset every destination row occurs at most once and the tile is private to the wave, so the
//    plain read-modify-write is exact; ds_add_f32 atomics measured ~10x slower.)
//  * The vector L1 spends ~4 (L1 hit) to ~7 (L2 hit) cycles per distinct 128-byte line a wave instruction touches:
//    a load instruction therefore reads 64/PPR COMPLETE rows (PPR = JB*4 lanes x 16 B each), and the wave
//    transposes them to MFMA fragments through a private XOR-swizzled LDS image (ds_write_b128 -> ds_read_b128,
//    both conflict-free; LDS operations of one wave execute in order, so no barrier).
template <int NI, int JB, int NW = 2>
struct GmmBuf {
    f32x4 a[NI];         // instruction i: rows i*RPI + lane/PPR of the item, 16-byte piece lane%PPR of the unit
    f32x4 b[JB][NW];     // weight fragments: fp32 [16-channel group][column block], bf16 [32-channel group][column block], x3 [(group, block)][plane]
};

struct GmmItem {
    int k, base, e;      // wave-uniform; when !valid the fields still name a real item (addresses stay legal)
    bool valid;
};

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// BF: bf16 MFMA operands (BASELINE configs[2], "MFMA bf16 on rule GEMM").  Rows are gathered as fp32 exactly as below and
// rounded to bf16 (RNE) only when the MFMA operand is formed: two 16-channel pieces of a unit make one
// v_mfma_f32_16x16x32_bf16 (k = 8q + e <-> channel 16 (2 jl + (e >> 2)) + 4q + (e & 3) for lane group q), the weight
// fragments come pre-rounded in that order from u3d_weight_pack_bf16 (half the weight bytes); accumulation, the LDS
// accumulator tile and the output stay fp32.  8 fp32 MFMAs (256 SIMD cycles) become one bf16 MFMA (16) + 4 v_cvt_pk.
// PR = 0: fp32 MFMAs; 1: bf16 operands (above); 2: fp32 products from three exact bf16 planes per operand (u3d_common.h "bf16x3"):
// same operand shapes as PR = 1, the gathered rows are split where PR = 1 rounds them, the weights come pre-split from
// u3d_weight_pack_x3 (three 1 KB blocks per (32-channel group, column block)), and a fragment pair takes six MFMAs.
template <int CS16, int R, int JB_, int PR = 0>
struct GmmWave {
    static constexpr bool BF = PR == 1, X3 = PR == 2;
    static constexpr int NCH = R / 32;            // 16-pair chunks per item
    static constexpr int JB = JB_;                // 16-channel groups per unit (1, 2 or 4)
    static constexpr int NJB = CS16 / JB;
    static constexpr int W = 16 * NCH;
    static constexpr int PPR = JB * 4;            // 16-byte pieces (lanes) per row of a unit
    static constexpr int RPI = 64 / PPR;          // rows per load instruction
    static constexpr int NI = W / RPI;            // load instructions per unit
    static constexpr int IPC = 16 / RPI;          // load instructions per 16-row chunk
    static constexpr int TRASH = R;               // scratch accumulator row for lanes past the end of a range
    static_assert(CS16 % JB == 0 && (JB == 1 || JB == 2 || JB == 4), "unit shape");
    using Buf = GmmBuf<NI, JB, X3 ? 3 : 2>;

    __amdgpu_buffer_rsrc_t rs_src, rs_g, rs_s, rs_w;
    char* accq;                                   // accumulator tile + this lane's column offset (q*16 bytes)
    float* stage;                                 // 16 rows x PPR pieces, piece slot p ^ swz(row)
    int lane, i16, slice, k_hi, cs4, K, row0;
    int64_t cap;
    int ts_s, ts_e;                               // lane k: pair range of offset k in this row tile
    int lr, lp16, lane16, lw, lw4, q16;           // per-lane constants
    int cg[1], cs_[1];                            // byte offsets of this lane's first entries in an item's index window (see load_idx)
    int wr_off, rd_off[JB];                       // float offsets into `stage` of this lane's write / fragment reads

    GmmItem it0, it1;                             // item being computed / item whose first unit is fetched next
    // Pair indices per lane, loaded straight into the lanes that use them (round 3; rounds 1-2 loaded one coalesced window and
    // redistributed it with NI + NCH ds_bpermute per item and per unit -- 10 us of a 145 us level-1 launch by ablation):
    int g_cur[NI];                                // it0: gather rows of the pairs this lane loads (pair i*RPI + lane/PPR of the window)
    int ix1_g[NI], ix1_s[NCH];                    // it1: gather rows; scatter rows of pair c*16 + lane%16 (this lane's MFMA column)
    int soff0, soff1;                             // it0: byte offset of this lane's accumulator row, chunk 0 / 1
    f32x4 d00, d01, d10, d11;                     // accumulators [chunk][column block]; named scalars: arrays get merged into
                                                  // runtime-indexed scratch by the TWO / single-chunk tail merge

    static __device__ __forceinline__ int swz(int row) { return PPR == 4 ? ((row >> 1) & 3) : (row & (PPR - 1)); }

    __device__ __forceinline__ void init(const GmmParams& p, float* acc, float* stage_, int lane_, int slice_, int64_t row0_) {
        // exact extents: an index window may run past the end of its range (entries of the next tile, uninitialised memory past
        // the offset's count, or -- past the list -- zeros) -- such lanes compute into the scratch row, and whatever row index
        // they read, the buffer bounds keep the gather inside `src` (out-of-range loads return 0)
        rs_src = make_rsrc(p.src, p.n_src * p.Cs * 4); rs_g = make_rsrc(p.gather, (int64_t)p.K * p.cap * 4);
        rs_s = make_rsrc(p.scatter, (int64_t)p.K * p.cap * 4); rs_w = make_rsrc(p.w);
        lane = lane_; i16 = lane & 15; slice = slice_; cs4 = p.Cs * 4; K = p.K; cap = p.cap; row0 = (int)row0_;
        const int q = lane >> 4;
        accq = reinterpret_cast<char*>(acc) + (GMM_SWZ ? 0 : q * 16);
        q16 = q << 4;
        stage = stage_;
        lr = lane / PPR; lp16 = (lane % PPR) * 16; lane16 = lane * 16; lw = lane & (W - 1); lw4 = lw * 4;
        cg[0] = lr * 4; cs_[0] = i16 * 4;
        wr_off = (lr * PPR + ((lane % PPR) ^ swz(lr))) * 4;             // swz(i*RPI + lr) == swz(lr) for PPR = 4, 8
#pragma unroll
        for (int j = 0; j < JB; ++j) rd_off[j] = (i16 * PPR + ((j * 4 + q) ^ swz(i16))) * 4;
    }
    __device__ __forceinline__ bool range_of(int k, int& s_, int& e_) const {
        s_ = __builtin_amdgcn_readlane(ts_s, k);
        e_ = __builtin_amdgcn_readlane(ts_e, k);
        return s_ < e_;
    }
    __device__ __forceinline__ GmmItem first_item(int k_lo) const {
        GmmItem n{k_lo, 0, 0, false};
        for (int k = k_lo; k < k_hi; ++k)
            if (range_of(k, n.base, n.e)) { n.k = k; n.valid = true; break; }
        return n;
    }
    __device__ __forceinline__ GmmItem next_of(const GmmItem& it) const {
        GmmItem n = it;
        if (!it.valid) return n;
        n.base = it.base + W;
        if (n.base < n.e) return n;
        for (int k = it.k + 1; k < k_hi; ++k) {
            int s_, e_;
            if (range_of(k, s_, e_)) { n.k = k; n.base = s_; n.e = e_; return n; }
        }
        n = it;
        n.valid = false;
        return n;
    }
    // NI + NCH dword loads per item: every lane fetches exactly the entries it will use -- its gather rows (the rows its load
    // instructions fetch) and its scatter rows (its MFMA column).  Two VALU adds per item (window start + lane part in the VGPR
    // offset, the lists' start k * cap in the scalar offset, the per-instruction part as immediate).  The descriptors span the
    // whole [K][cap] index array (the hardware's range check covers scalar + vector + immediate offset -- measured: a
    // descriptor of one list with the list start in the scalar offset reads zeros for every k >= 1): a window that runs past
    // its list reads entries of the next offset's list (valid rows), past the array it reads 0.
    __device__ __forceinline__ void load_idx(const GmmItem& it, int (&g)[NI], int (&s_)[NCH]) const {
        const int soff_k = (int)(it.k * cap) * 4;
        const int vg = cg[0] + it.base * 4, vs = cs_[0] + it.base * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i) g[i] = bload32(rs_g, vg + i * (RPI * 4), soff_k);
#pragma unroll
        for (int c = 0; c < NCH; ++c) s_[c] = bload32(rs_s, vs + c * 64, soff_k);
    }
    __device__ __forceinline__ void issue(Buf& buf, const int (&g)[NI], int k, int u) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) buf.a[i] = bload128(rs_src, (int)__umul24(g[i], cs4) + lp16, u * (JB * 64));
        if constexpr (X3) {      // 32-channel groups: (jl, nb, plane) blocks of 1 KB (8 bf16 per lane)
            const int wso = ((slice * K + k) * (CS16 / 2) + u * (JB / 2)) * 6144;
#pragma unroll
            for (int j = 0; j < JB / 2; ++j)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int q = 0; q < 3; ++q) buf.b[j * 2 + nb][q] = bload128(rs_w, lane16 + (nb * 3 + q) * 1024, wso + j * 6144);
        } else if constexpr (BF) {      // 32-channel groups: (jl, nb) blocks of 1 KB (8 bf16 per lane)
            const int wso = ((slice * K + k) * (CS16 / 2) + u * (JB / 2)) * 2048;
#pragma unroll
            for (int j = 0; j < JB / 2; ++j) {
                buf.b[j][0] = bload128(rs_w, lane16, wso + j * 2048);
                buf.b[j][1] = bload128(rs_w, lane16 + 1024, wso + j * 2048);
            }
        } else if constexpr ((U3D_GMM_ABL & 32) != 0) {
#pragma unroll
            for (int j = 0; j < JB; ++j) { asm volatile("" : "+v"(buf.b[j][0]), "+v"(buf.b[j][1])); }
        } else {
            const int wso = ((slice * K + k) * CS16 + u * JB) * 2048;       // bytes: (j, nb) blocks of 1 KB
#pragma unroll
            for (int j = 0; j < JB; ++j) {
                buf.b[j][0] = bload128(rs_w, lane16, wso + j * 2048);
                buf.b[j][1] = bload128(rs_w, lane16 + 1024, wso + j * 2048);
            }
        }
    }
    static __device__ __forceinline__ bf16x8 cvt8(const f32x4& lo, const f32x4& hi) {
        return bf16x8{(__bf16)lo[0], (__bf16)lo[1], (__bf16)lo[2], (__bf16)lo[3], (__bf16)hi[0], (__bf16)hi[1], (__bf16)hi[2], (__bf16)hi[3]};
    }
    struct Frag { f32x4 v[JB]; };
    // rows of one 16-pair chunk: registers -> swizzled LDS image -> fragments
    template <int C>
    __device__ __forceinline__ Frag frags(const Buf& buf) const {
        if constexpr ((U3D_GMM_ABL & 8) != 0) {
            Frag f;
#pragma unroll
            for (int j = 0; j < JB; ++j) f.v[j] = buf.a[(C * IPC + j) % NI];
            return f;
        }
#pragma unroll
        for (int i = 0; i < IPC; ++i) {
            int off = wr_off + i * (RPI * PPR * 4);
            if constexpr (PPR == 16) {       // swz(row) depends on i here: recompute the slot
                const int row = i * RPI + lr;
                off = (row * PPR + ((lp16 >> 4) ^ swz(row))) * 4;
            }
            *reinterpret_cast<f32x4*>(stage + off) = buf.a[C * IPC + i];
        }
        Frag f;
#pragma unroll
        for (int j = 0; j < JB; ++j) f.v[j] = *reinterpret_cast<const f32x4*>(stage + rd_off[j]);
        return f;
    }

    // byte offsets of this lane's accumulator rows (chunk 0 / 1) of item `it` from the scatter rows it loaded
    __device__ __forceinline__ void row_offsets(const GmmItem& it, const int (&s_)[NCH], int& o0, int& o1) const {
        const int left = it.e - it.base;                   // pairs of the window that exist
        const int r0 = i16 < left ? s_[0] - row0 : TRASH;
        const int r1 = NCH == 2 ? (16 + i16 < left ? s_[NCH - 1] - row0 : TRASH) : 0;
        if constexpr (GMM_SWZ) {       // byte offset of quad 0 of the row, then this lane's quad q: (q ^ (row & 7)) << 4
            o0 = ((r0 << 7) | ((r0 & 7) << 4)) ^ q16;
            o1 = NCH == 2 ? ((r1 << 7) | ((r1 & 7) << 4)) ^ q16 : 0;
        } else {
            o0 = (int)__umul24(r0, GMM_ALD * 4);
            o1 = NCH == 2 ? (int)__umul24(r1, GMM_ALD * 4) : 0;
        }
    }
    // byte offset of columns 16..31 (quad q + 4) of the row whose columns 0..15 sit at byte offset o
    static __device__ __forceinline__ int hi_cols(int o) { return GMM_SWZ ? (o ^ 64) : (o + 64); }

    // One unit of the current item.  Chunk 1 (pairs 16..31 of the window) exists only when the window holds more than
    // 16 pairs (wave-uniform).  Order matters: a wave's critical path per item is LDS round trips + MFMAs, so ALL LDS
    // work of the unit (accumulator reads, staging writes, fragment reads, index shuffles for the next unit) is put
    // into the in-order LDS queue up front and waited for once, then the next unit's loads are issued and the MFMAs
    // run back to back.
    template <int U>
    __device__ __forceinline__ void unit(Buf& cur, Buf& nxt) {
        const bool two = NCH == 2 && it0.base + 16 < it0.e;
        if constexpr (U == 0 && (U3D_GMM_ABL & 2) != 0) {
            d00 = f32x4{0.f, 0.f, 0.f, 0.f}; d01 = d00; d10 = d00; d11 = d00;
        } else if constexpr (U == 0) {         // accumulator rows of the item -> C operands
            d00 = *reinterpret_cast<const f32x4*>(accq + soff0);
            d01 = *reinterpret_cast<const f32x4*>(accq + hi_cols(soff0));
            if (two) {
                d10 = *reinterpret_cast<const f32x4*>(accq + soff1);
                d11 = *reinterpret_cast<const f32x4*>(accq + hi_cols(soff1));
            }
        }
        const Frag f0 = frags<0>(cur);
        Frag f1 = f0;
        if (two) f1 = frags<NCH - 1>(cur);
        GmmItem it2;
        int g2[NI], s2[NCH], n0 = 0, n1 = 0;
        if constexpr (U == NJB - 1) {
            row_offsets(it1, ix1_s, n0, n1);
            it2 = next_of(it1);
            load_idx(it2, g2, s2);
            issue(nxt, ix1_g, it1.k, 0);
        } else {
            issue(nxt, g_cur, it0.k, U + 1);
        }
        if constexpr (X3) {
            // The h.h product of a fragment pair accumulates straight into the running row (the MFMA's C operand), the five
            // low-order plane products into a ZERO-initialised accumulator that the VALU adds at the end of the unit: the bf16
            // MFMA aligns its 32 products to the exponent of its C operand and truncates them there, always towards zero
            // (tools/bias_probe.py), so products 2^-8 .. 2^-16 below a running sum lose low bits on every instruction -- a
            // coherent bias that sums over hundreds of thousands of rows (weight gradients, batch-norm statistics) do not
            // average out: 4x larger gradient errors end to end.  Against their own sum they lose nothing that matters.
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            f32x4 t00 = z, t01 = z, t10 = z, t11 = z;
#pragma unroll
            for (int j = 0; j < JB / 2; ++j) {
                bf16x8 x0[3];
                split3_x8(f0.v[2 * j], f0.v[2 * j + 1], x0);
#pragma unroll
                for (int o = 2; o >= 1; --o)
#pragma unroll
                    for (int qa = 0; qa <= o; ++qa) {
                        t00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur.b[j * 2][qa]), x0[o - qa], t00, 0, 0, 0);
                        t01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur.b[j * 2 + 1][qa]), x0[o - qa], t01, 0, 0, 0);
                    }
                d00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur.b[j * 2][0]), x0[0], d00, 0, 0, 0);
                d01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur.b[j * 2 + 1][0]), x0[0], d01, 0, 0, 0);
            }
            if (two) {
#pragma unroll
                for (int j = 0; j < JB / 2; ++j) {
                    bf16x8 x1[3];
                    split3_x8(f1.v[2 * j], f1.v[2 * j + 1], x1);
#pragma unroll
                    for (int o = 2; o >= 1; --o)
#pragma unroll
                        for (int qa = 0; qa <= o; ++qa) {
                            t10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur.b[j * 2][qa]), x1[o - qa], t10, 0, 0, 0);
                            t11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur.b[j * 2 + 1][qa]), x1[o - qa], t11, 0, 0, 0);
                        }
                    d10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur.b[j * 2][0]), x1[0], d10, 0, 0, 0);
                    d11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur.b[j * 2 + 1][0]), x1[0], d11, 0, 0, 0);
                }
                d10 += t10; d11 += t11;
            }
            d00 += t00; d01 += t01;
        } else if constexpr (BF) {
#pragma unroll
            for (int j = 0; j < JB / 2; ++j) {
                const bf16x8 x0 = cvt8(f0.v[2 * j], f0.v[2 * j + 1]);
                d00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur.b[j][0]), x0, d00, 0, 0, 0);
                d01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur.b[j][1]), x0, d01, 0, 0, 0);
            }
            if (two) {
#pragma unroll
                for (int j = 0; j < JB / 2; ++j) {
                    const bf16x8 x1 = cvt8(f1.v[2 * j], f1.v[2 * j + 1]);
                    d10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur.b[j][0]), x1, d10, 0, 0, 0);
                    d11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur.b[j][1]), x1, d11, 0, 0, 0);
                }
            }
        } else if constexpr ((U3D_GMM_ABL & 1) != 0) {
#pragma unroll
            for (int j = 0; j < JB; ++j) asm volatile("" :: "v"(cur.b[j][0]), "v"(cur.b[j][1]), "v"(f0.v[j]), "v"(f1.v[j]));
        } else {
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                d00 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.b[j][0][t], f0.v[j][t], d00, 0, 0, 0);
                d01 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.b[j][1][t], f0.v[j][t], d01, 0, 0, 0);
            }
        if (two) {
#pragma unroll
            for (int j = 0; j < JB; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    d10 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.b[j][0][t], f1.v[j][t], d10, 0, 0, 0);
                    d11 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.b[j][1][t], f1.v[j][t], d11, 0, 0, 0);
                }
        }
        }
        if constexpr (U == NJB - 1 && (U3D_GMM_ABL & 2) != 0) {
            asm volatile("" :: "v"(d00), "v"(d01), "v"(d10), "v"(d11));
            rotate(g2, s2); soff0 = n0; soff1 = n1;
            it0 = it1; it1 = it2;
        } else if constexpr (U == NJB - 1) {
            *reinterpret_cast<f32x4*>(accq + soff0) = d00;
            *reinterpret_cast<f32x4*>(accq + hi_cols(soff0)) = d01;
            if (two) {
                *reinterpret_cast<f32x4*>(accq + soff1) = d10;
                *reinterpret_cast<f32x4*>(accq + hi_cols(soff1)) = d11;
            }
            rotate(g2, s2); soff0 = n0; soff1 = n1;
            it0 = it1; it1 = it2;
        }
    }

    __device__ __forceinline__ void rotate(const int (&g2)[NI], const int (&s2)[NCH]) {
#pragma unroll
        for (int i = 0; i < NI; ++i) { g_cur[i] = ix1_g[i]; ix1_g[i] = g2[i]; }
#pragma unroll
        for (int c = 0; c < NCH; ++c) ix1_s[c] = s2[c];
    }

    // units U .. NJB-1 of the current item, buffers alternating
    template <int U>
    __device__ __forceinline__ void units(Buf& a, Buf& b) {
        unit<U>(a, b);
        if constexpr (U + 1 < NJB) units<U + 1>(b, a);
    }

    __device__ __forceinline__ void run(int k_lo) {
        it0 = first_item(k_lo);
        if (!it0.valid) return;
        int s_first[NCH];
        load_idx(it0, g_cur, s_first);
        it1 = next_of(it0);
        load_idx(it1, ix1_g, ix1_s);
        row_offsets(it0, s_first, soff0, soff1);
        Buf X, Y;
        issue(X, g_cur, it0.k, 0);
        while (true) {
            units<0>(X, Y);
            if (!it0.valid) break;
            if constexpr (NJB % 2 == 1) {          // an odd unit count leaves the next item's first unit in Y
                units<0>(Y, X);
                if (!it0.valid) break;
            }
        }
    }
};

// 16-channel groups per unit: 128-byte row pieces (JB = 2) for 64-row tiles, 256-byte pieces for 32-row tiles when
// the channel count allows
constexpr int gmm_jb(int cs16, int r) { return (cs16 % 4 == 0 && r == 32) ? 4 : (cs16 % 2 == 0 ? 2 : 1); }
// floats: accumulator rows (+ the scratch row of the padded layout; the swizzled layout's scratch row IS the first 128 bytes
// of the staging image) + staging image
constexpr int gmm_acc_rows(int r) { return GMM_SWZ ? r : r + 1; }
constexpr int gmm_wave_lds(int cs16, int r) { return gmm_acc_rows(r) * GMM_ALD + 16 * gmm_jb(cs16, r) * 16; }

template <int CS16, int R, int PR = 0>
__global__ __launch_bounds__(256) void spconv_gmm_k(GmmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: everything derived from it stays in SGPRs
    float* acc = smem + wave * gmm_wave_lds(CS16, R);                // R rows (+ scratch row), then the staging image

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
        *reinterpret_cast<float4*>(acc + gmm_acc_idx(r, c4)) = v;
    }

    static_assert(PR == 0 || CS16 % 2 == 0, "bf16 operands pair 16-channel groups");
    GmmWave<CS16, R, gmm_jb(CS16, R), PR> w;
    w.init(p, acc, acc + gmm_acc_rows(R) * GMM_ALD, lane, slice, row0);
    const int k_lo = g * p.kper;
    w.k_hi = min(p.K, k_lo + p.kper);
    // all (start, end) ranges of this wave's offsets in one round trip: lane k holds offset k's range
    w.ts_s = 0; w.ts_e = 0;
    if (lane < p.K) {
        w.ts_s = p.ts[lane * tsld + sub];
        w.ts_e = p.ts[lane * tsld + sub + 1];
    }
    w.run(k_lo);

    float* out = p.out + (p.G > 1 ? (int64_t)g * p.n_dst * p.Cd : 0);
    // dst rows leave the LDS tile here; the batch norm that follows (every convolution of the U-Net feeds one) needs sum x and
    // sum x^2 per channel: accumulated on the way out (lane = column quad c4, rows lane/8 + 8 i), folded over the 8 row groups
    // with three shuffles and stored as ONE partial row per tile -- bn_partials_k adds the tiles in fp64.  This replaces a
    // pass over dst (bn_reduce_k<0>) per layer; ~60 VALU instructions per tile, outside the MFMA loop.
    if (!p.stats) {                                    // wave-uniform (input-gradient launches, eval mode)
        for (int idx = lane; idx < rows * (GMM_CDS / 4); idx += 64) {
            const int r = idx >> 3, c4 = idx & 7;
            *reinterpret_cast<float4*>(out + (row0 + r) * p.Cd + n0 + c4 * 4) = *reinterpret_cast<const float4*>(acc + gmm_acc_idx(r, c4));
        }
        return;
    }
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
    for (int idx = lane; idx < rows * (GMM_CDS / 4); idx += 64) {
        const int r = idx >> 3, c4 = idx & 7;
        const float4 v = *reinterpret_cast<const float4*>(acc + gmm_acc_idx(r, c4));
        *reinterpret_cast<float4*>(out + (row0 + r) * p.Cd + n0 + c4 * 4) = v;
        s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
        s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
    }
    {
#pragma unroll
        for (int d = 8; d <= 32; d <<= 1) {
            s1.x += __shfl_xor(s1.x, d, 64); s1.y += __shfl_xor(s1.y, d, 64); s1.z += __shfl_xor(s1.z, d, 64); s1.w += __shfl_xor(s1.w, d, 64);
            s2.x += __shfl_xor(s2.x, d, 64); s2.y += __shfl_xor(s2.y, d, 64); s2.z += __shfl_xor(s2.z, d, 64); s2.w += __shfl_xor(s2.w, d, 64);
        }
        if (lane < 8) {
            float* st = p.stats + sub * 2 * p.Cd + n0 + lane * 4;
            *reinterpret_cast<float4*>(st) = s1;
            *reinterpret_cast<float4*>(st + p.Cd) = s2;
        }
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

// rows per wave-tile and offset groups.  Measured on MI355X at the cfg2 level sizes (tools/prof_gmm.py sweep, round 4,
// profiles/round4_gmm_plan_sweep.txt): 64-row tiles win at every level once a level that cannot fill the chip with them
// (fewer than 2048 wave tiles) splits its 27 offsets over NINE groups instead of going to 32-row tiles and three groups
// (level 3, 192 -> 96 channels: 126 us against 161; level 4, 256 -> 128: 69 against 80); strided / inverse convolutions
// (8 offsets, no groups) keep 32-row tiles there for the parallelism.
// The bf16-row kernel (u3d_spconv_gmm_bf16a: light items, occupancy bound by the accumulator tile) wants the opposite: 32-row tiles,
// and offset groups only as far as it takes to reach ~2048 wave tiles (same sweep on that kernel,
// profiles/round4_gmm_plan_sweep_bf16rows.txt: level 3, 96 -> 96: 35 us against 42; level 4, 128 -> 128: 25 against 31).
static void plan_gmm(int Cs, int Cd, int K, int64_t n_dst, int* R, int* G, bool rows_kernel = false) {
    const int slices = Cd / GMM_CDS;
    const int64_t want = 2048;
    int r = 64, g = 1;
    if (rows_kernel) {            // 32-row tiles at EVERY level (level 1, 32 -> 32: 61 us against 67; level 2, 64 -> 64: 54 against 64): 21 KB of LDS
        r = 32;                   // per workgroup instead of 37, seven workgroups per CU
        const int64_t waves = ceil_div(n_dst, 32) * slices;
        if (waves < want && K >= 27) g = waves * 3 >= want ? 3 : 9;
    } else if (ceil_div(n_dst, 64) * slices < want) {
        if (K >= 27) g = 9;
        else r = 32;
    }
    if (const char* e = getenv("U3D_GMM_R")) r = atoi(e) == 32 ? 32 : 64;      // experiment knobs (tools/prof_gmm.py)
    if (const char* e = getenv("U3D_GMM_G")) { const int v = atoi(e); if ((v == 1 || v == 3 || v == 9) && K >= 27) g = v; }
    *R = r;
    *G = g;
}

template <int CS16, int R, int PR = 0>
static int launch_gmm(const GmmParams& p, hipStream_t s) {
    const size_t lds = (size_t)4 * gmm_wave_lds(CS16, R) * sizeof(float);
    const int64_t waves = p.n_sub * p.n_slices * p.G;
    hipLaunchKernelGGL((spconv_gmm_k<CS16, R, PR>), dim3((unsigned)ceil_div(waves, 4)), dim3(256), lds, s, p);
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

// N consecutive floats of a row through raw buffer loads at byte offset voff: 16-byte loads where the strided channel
// blocks are 16-byte aligned, dwords otherwise (this compiler lowers __builtin_amdgcn_raw_buffer_load_b64 to a ONE-dword
// load -- checked in the ISA -- so there is no 8-byte form)
template <int N>
__device__ __forceinline__ void load_row_part(float (&v)[N], __amdgpu_buffer_rsrc_t r, int voff) {
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (int s = 0; s < N; s += 4) {
            const f32x4 t = bload128(r, voff + s * 4, 0);
            v[s] = t[0]; v[s + 1] = t[1]; v[s + 2] = t[2]; v[s + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int s = 0; s < N; ++s) v[s] = __builtin_bit_cast(float, bload32(r, voff + s * 4, 0));
    }
}

// NG = Cd/16, NX = Cs/16 floats per lane per row; SG x SX waves share one pair range.
// BF (BASELINE configs[2]): the same walk with v_mfma_f32_16x16x32_bf16 -- 32 pairs per instruction, lane group q takes pairs
// 8q .. 8q+7 of a trip and rounds its row parts to bf16 (RNE) as it forms the operands; fp32 accumulation and partials.
// PR = 2 (u3d_common.h "bf16x3"): the BF walk with both row parts split exactly into three bf16 planes and six MFMAs per block
// pair -- fp32-level error; used only for the widest layers (see spconv_wgrad_impl for the measurements).
template <int NG, int NX, int SG, int SX, int PR = 0, bool PIPE = true>
__global__ __launch_bounds__(256) void spconv_wgrad_k(WgParams p) {
    constexpr bool BF = PR != 0, X3 = PR == 2;
    constexpr int NGW = NG / SG, NXW = NX / SX, SPLIT = SG * SX, RPW = 4 / SPLIT;   // RPW ranges per workgroup
    constexpr int CD = NG * 16, CS = NX * 16;
    // A range is the set of pairs of offset k whose dy row lies in row tile t (tile_starts), so the 27 offsets of
    // one tile read the SAME dy rows and neighbouring x rows.  Workgroup b runs on XCD b % 8 (private 4 MB L2):
    // XCD x takes tile groups x, x+8, x+16, ... and walks ALL offsets of a group back to back, so those rows are
    // fetched once and re-read from the XCD's own L2 (PMC with pair-index ranges: TCC hit rate 11 %, 0.95 GB
    // fetched per L1 launch for 125 MB of algorithmic traffic).
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int k = j % p.K;
    const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: the range bounds and every offset derived from them stay in
                                                                      // SGPRs (as a VGPR each index load became a waterfall loop)
    const int sub = wave % SPLIT, wa = sub % SG, wb = sub / SG;
    const int range = ((j / p.K) * 8 + xcd) * RPW + wave / SPLIT;
    // SPLIT == 1: the four waves of a workgroup walk four consecutive ranges of the same offset and add their accumulators
    // through LDS at the end (one partial block per workgroup: a quarter of the partial traffic and of the fixed-order
    // reduce's reads) -- such a wave stays for the two barriers even without a range of its own.
    constexpr bool LDSR = SPLIT == 1;
    const bool active = range < p.n_tiles;               // wave-uniform
    if (!LDSR && !active) return;
    const int lo = active ? p.ts[(int64_t)k * (p.n_tiles + 1) + range] : 0;
    const int hi = active ? p.ts[(int64_t)k * (p.n_tiles + 1) + range + 1] : 0;
    // VALU instructions do not overlap with fp32 MFMAs on a SIMD (tools/coissue.hip), so the loop is written for VALU
    // count: raw buffer loads (one v_mad_u32_u24 per gathered row instead of a 64-bit address chain), the pair list
    // read from a scalar offset with constant per-lane offsets (lane group q takes pairs 4q..4q+3 of the trip;
    // any fixed pair <-> MFMA k-slot assignment is valid as long as x and dy use the same), and no masking in full trips.
    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x), rs_g = make_rsrc(p.dy), rs_rx = make_rsrc(p.rows_x), rs_rg = make_rsrc(p.rows_dy);
    const int ksoff = (int)(k * p.cap) * 4;
    const int gvo = (NG * i16 + wa * NGW) * 4, xvo = (NX * i16 + wb * NXW) * 4;     // byte offset of this lane's channel block in a row
    const int q16 = q * 16;

    f32x4 acc[NGW][NXW];
#pragma unroll
    for (int a = 0; a < NGW; ++a)
#pragma unroll
        for (int b = 0; b < NXW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    int base = lo;
    if constexpr (BF) {
        const int q32 = q * 32;
        for (; base < hi; base += 32) {            // 32 pairs per trip; the last trip clamps its indices into the range and masks dy
            const bool full = base + 32 <= hi;       // wave-uniform
            float gv[8][NGW], xv[8][NXW];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int io, ix;
                if (full) {
                    io = bload32(rs_rg, q32 + u * 4, ksoff + base * 4);
                    ix = bload32(rs_rx, q32 + u * 4, ksoff + base * 4);
                } else {
                    const int pi = min(base + 8 * q + u, hi - 1);
                    io = bload32(rs_rg, pi * 4, ksoff);
                    ix = bload32(rs_rx, pi * 4, ksoff);
                }
                load_row_part<NGW>(gv[u], rs_g, (int)__umul24(io, CD * 4) + gvo);
                load_row_part<NXW>(xv[u], rs_x, (int)__umul24(ix, CS * 4) + xvo);
                if (!full && base + 8 * q + u >= hi) {          // pairs past the end contribute exact zeros
#pragma unroll
                    for (int a = 0; a < NGW; ++a) gv[u][a] = 0.f;
                }
            }
            if constexpr (X3) {
                bf16x8 xb[NXW][3];
#pragma unroll
                for (int b = 0; b < NXW; ++b)
                    split3_x8(f32x4{xv[0][b], xv[1][b], xv[2][b], xv[3][b]}, f32x4{xv[4][b], xv[5][b], xv[6][b], xv[7][b]}, xb[b]);
#pragma unroll
                for (int a = 0; a < NGW; ++a) {
                    bf16x8 ga[3];
                    split3_x8(f32x4{gv[0][a], gv[1][a], gv[2][a], gv[3][a]}, f32x4{gv[4][a], gv[5][a], gv[6][a], gv[7][a]}, ga);
#pragma unroll
                    for (int o = 2; o >= 0; --o)
#pragma unroll
                        for (int qa = 0; qa <= o; ++qa)
#pragma unroll
                            for (int b = 0; b < NXW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga[qa], xb[b][o - qa], acc[a][b], 0, 0, 0);
                }
                continue;
            }
            bf16x8 xb[NXW];
#pragma unroll
            for (int b = 0; b < NXW; ++b)
                xb[b] = bf16x8{(__bf16)xv[0][b], (__bf16)xv[1][b], (__bf16)xv[2][b], (__bf16)xv[3][b], (__bf16)xv[4][b], (__bf16)xv[5][b], (__bf16)xv[6][b], (__bf16)xv[7][b]};
#pragma unroll
            for (int a = 0; a < NGW; ++a) {
                const bf16x8 ga = bf16x8{(__bf16)gv[0][a], (__bf16)gv[1][a], (__bf16)gv[2][a], (__bf16)gv[3][a], (__bf16)gv[4][a], (__bf16)gv[5][a], (__bf16)gv[6][a], (__bf16)gv[7][a]};
#pragma unroll
                for (int b = 0; b < NXW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ga, xb[b], acc[a][b], 0, 0, 0);
            }
        }
    }
    if constexpr (!PIPE) {
        for (; base + 16 <= hi; base += 16) {        // full trips: 4 MFMA K-steps (16 pairs), all loads up front
            int io[4], ix[4];        // dword loads: `base` is only 4-byte aligned and 16-byte buffer loads are size-aligned by the hardware
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                io[u] = bload32(rs_rg, q16 + u * 4, ksoff + base * 4);
                ix[u] = bload32(rs_rx, q16 + u * 4, ksoff + base * 4);
            }
            float gv[4][NGW], xv[4][NXW];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                load_row_part<NGW>(gv[u], rs_g, (int)__umul24(io[u], CD * 4) + gvo);
                load_row_part<NXW>(xv[u], rs_x, (int)__umul24(ix[u], CS * 4) + xvo);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int a = 0; a < NGW; ++a)
#pragma unroll
                    for (int b = 0; b < NXW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[u][a], xv[u][b], acc[a][b], 0, 0, 0);
        }
    }
    // full trips (16 pairs = 4 MFMA K-steps), software-pipelined three deep: while the MFMAs of trip t run, the rows of trip t+1
    // and the pair indices of trip t+2 are in flight (a wave's gather is a chain of two dependent round trips, index -> row).
    // Every iteration issues the same loads (the tail re-reads the last full trip instead of branching) so the wait counters
    // stay exact; two register sets alternate, the loop is unrolled by two.  Measured: 4.67 -> 4.54 ms/step over all levels.
    const int nfull = PIPE ? (hi - base) >> 4 : 0;
    if (nfull > 0) {
        const int last = base + (nfull - 1) * 16;
        int ioA[4], ixA[4], ioB[4], ixB[4];
        float gA[4][NGW], xA[4][NXW], gB[4][NGW], xB[4][NXW];
        auto load_idx = [&](int (&io)[4], int (&ix)[4], int b) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                io[u] = bload32(rs_rg, q16 + u * 4, ksoff + b * 4);
                ix[u] = bload32(rs_rx, q16 + u * 4, ksoff + b * 4);
            }
        };
        auto load_rows = [&](float (&gv)[4][NGW], float (&xv)[4][NXW], const int (&io)[4], const int (&ix)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {       // (U3D_WG_ABL == 3: timing ablation, every gather hits rows 0..63)
                load_row_part<NGW>(gv[u], rs_g, (int)__umul24(U3D_WG_ABL == 3 ? (io[u] & 63) : io[u], CD * 4) + gvo);
                load_row_part<NXW>(xv[u], rs_x, (int)__umul24(U3D_WG_ABL == 3 ? (ix[u] & 63) : ix[u], CS * 4) + xvo);
            }
        };
        auto mfmas = [&](const float (&gv)[4][NGW], const float (&xv)[4][NXW]) {
            if constexpr (U3D_WG_ABL == 1) {    // timing ablation: operands kept alive, no MFMA
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int a = 0; a < NGW; ++a) asm volatile("" :: "v"(gv[u][a]));
#pragma unroll
                    for (int b = 0; b < NXW; ++b) asm volatile("" :: "v"(xv[u][b]));
                }
                return;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int a = 0; a < NGW; ++a)
#pragma unroll
                    for (int b = 0; b < NXW; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[u][a], xv[u][b], acc[a][b], 0, 0, 0);
        };
        load_idx(ioA, ixA, base);
        load_idx(ioB, ixB, min(base + 16, last));
        load_rows(gA, xA, ioA, ixA);
        for (int t = 0; t < nfull; t += 2) {
            load_rows(gB, xB, ioB, ixB);                               // rows of trip t+1
            load_idx(ioA, ixA, min(base + (t + 2) * 16, last));        // indices of trip t+2
            mfmas(gA, xA);                                             // trip t
            if (t + 1 >= nfull) break;
            load_rows(gA, xA, ioA, ixA);                               // rows of trip t+2
            load_idx(ioB, ixB, min(base + (t + 3) * 16, last));        // indices of trip t+3
            mfmas(gB, xB);                                             // trip t+1
        }
        base += nfull * 16;
    }
    if (base < hi) {                              // last, partial trip: indices clamped into the range, masked dy
        float gv[4][NGW], xv[4][NXW];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pi = min(base + 4 * q + u, hi - 1);
            const int io = bload32(rs_rg, pi * 4, ksoff), ix = bload32(rs_rx, pi * 4, ksoff);
            load_row_part<NGW>(gv[u], rs_g, (int)__umul24(io, CD * 4) + gvo);
            load_row_part<NXW>(xv[u], rs_x, (int)__umul24(ix, CS * 4) + xvo);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = base + 4 * q + u < hi;        // pairs past the end contribute exact zeros
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
    if constexpr (LDSR) {
        // ((w0 + w2) + (w1 + w3)): fixed order -> deterministic.  Accumulator element (a, b, r) of lane l sits at [(a NXW + b) 4 + r][l].
        constexpr int EW = NGW * NXW * 256;
        __shared__ float red[2][EW];
        float* mine = red[wave & 1] + lane;
        if (wave < 2) {
#pragma unroll
            for (int a = 0; a < NGW; ++a)
#pragma unroll
                for (int b = 0; b < NXW; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[((a * NXW + b) * 4 + r) * 64] = acc[a][b][r];
        }
        __syncthreads();
        if (wave >= 2) {
#pragma unroll
            for (int a = 0; a < NGW; ++a)
#pragma unroll
                for (int b = 0; b < NXW; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mine[((a * NXW + b) * 4 + r) * 64] += acc[a][b][r];
        }
        __syncthreads();
        const int group = range >> 2;                    // = the workgroup's range group; at least its first range exists
        if ((range & ~3) >= p.n_tiles) return;
        float* out = p.partial + ((int64_t)k * ceil_div(p.n_tiles, 4) + group) * (CD * CS);
        for (int e = threadIdx.x; e < EW; e += 256) out[e] = red[0][e] + red[1][e];
        return;
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

// rows per tile: as many tiles as a 64 MB partial buffer, a cap of 256 and >= 64 rows per tile allow
static int plan_wgrad_rows(int K, int64_t n_rows, int Cs, int Cd) {
    static const int64_t cap = [] { const char* e = getenv("U3D_WGRAD_TILES"); const int64_t v = e ? atoll(e) : 0; return v > 0 ? v : (int64_t)256; }();
    static const int64_t budget = [] { const char* e = getenv("U3D_WGRAD_PARTIAL_MB"); const int64_t v = e ? atoll(e) : 0; return (v > 0 ? v : (int64_t)64) << 20; }();
    int64_t nt = budget / ((int64_t)K * Cs * Cd * 4);
    // 256 tiles, more for levels beyond ~330 k rows so that a tile stays near 1300 rows -- near 520 rows for the 27-offset 32 -> 32
    // layers of level 1, up to what the partial budget allows (606 tiles there).  Round 5 sweep on the pipelined + LDS-pre-reduced walk (profiles/round5_wgrad_tile_sweep.txt), level 1, 8 scenes
    // (356 k rows): 273 tiles 153 us, 384: 144, 512: 141, 606: 139, 767: 139, 1023: 146, 1534: 161; level 2 (81 k rows) keeps its 151
    // (256: 126 us, 512: 128, 768: 139 against 123), level 3 its 67 (254: 142 us against 85).  (Round 4, 16 scenes = 699 k rows at
    // level 1: 512 tiles 283 us against 313 at 256 for the fp32-row walk, 116 against 125 for the bf16-row kernel.)
    // (only the measured shape moves: the other level-1 layers -- 16 -> 32, 64 -> 32, the 8-offset strided / inverse pair -- keep 1300 rows)
    const int64_t rows_per = (K >= 27 && Cs == 32 && Cd == 32) ? 520 : 1300;
    const int64_t cap_n = cap > n_rows / rows_per ? cap : n_rows / rows_per;
    nt = nt > cap_n ? cap_n : (nt < 1 ? 1 : nt);
    const int64_t by_len = ceil_div(n_rows, 64);
    if (nt > by_len) nt = by_len;
    return (int)ceil_div(n_rows, nt);
}

template <int NX, int NG, int PR = 0>     // (Cs/16, Cd/16) as the dispatch macro passes them
static int launch_wgrad(const WgParams& p0, float* dW, hipStream_t s) {
    // split over the 4 waves until a wave's sub-block is <= 32 accumulators (128 VGPRs)
    constexpr int SG = (NG * NX > 32 && NG % 2 == 0 && NG >= NX) ? 2 : ((NG * NX > 64 && NG % 2 == 0) ? 2 : 1);
    constexpr int SX = ((NG / SG) * NX > 32 && NX % 2 == 0) ? 2 : 1;
    constexpr int SPLIT = SG * SX, RPW = 4 / SPLIT;
    static_assert(SPLIT == 1 || SPLIT == 2 || SPLIT == 4, "wave split");
    const WgParams& p = p0;
    const int64_t groups = ceil_div(ceil_div(p.n_tiles, RPW), 8) * 8;
    // software-pipelined walk unless its second register set would push the kernel below two waves per SIMD
    static const int pipe_env = [] { const char* e = getenv("U3D_WGRAD_PIPE"); return e ? atoi(e) : -1; }();
    const bool pipe = pipe_env >= 0 ? pipe_env != 0 : (NG / SG) * (NX / SX) <= 16;
    if (pipe) hipLaunchKernelGGL((spconv_wgrad_k<NG, NX, SG, SX, PR, true>), dim3((unsigned)(groups * p.K)), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((spconv_wgrad_k<NG, NX, SG, SX, PR, false>), dim3((unsigned)(groups * p.K)), dim3(256), 0, s, p);
    const int n_blocks = SPLIT == 1 ? (int)ceil_div(p.n_tiles, 4) : p.n_tiles;       // partial blocks per offset (see the LDS reduction in the kernel)
    hipLaunchKernelGGL(wgrad_reduce_k, dim3(NG * NX, p.K), dim3(256), 0, s, (const float*)p.partial, n_blocks, p.K, NG, NX, SG, SX, dW);
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

// bf16 form: one 16-byte vector (8 bf16) per (lane, 32-channel group jj, column block nb):
// wp[(((slice*K + k)*CS32 + jj)*2 + nb)*64 + lane][e] = bf16(W(n = slice*32 + nb*16 + (lane&15), k, c = (2 jj + (e>>2))*16 + (lane>>4)*4 + (e&3)))
__global__ __launch_bounds__(256) void weight_pack_bf16_k(const float* __restrict__ w, bf16x8* __restrict__ wp, int Cd, int K, int Cs, int transposed) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one bf16x8 per thread
    const int cs32 = Cs / 32;
    const int64_t total = (int64_t)(Cd / 32) * K * cs32 * 2 * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    int64_t t = idx >> 6;
    const int nb = (int)(t & 1); t >>= 1;
    const int jj = (int)(t % cs32); t /= cs32;
    const int k = (int)(t % K);
    const int slice = (int)(t / K);
    const int n = slice * 32 + nb * 16 + (lane & 15), q = lane >> 4;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = (2 * jj + (e >> 2)) * 16 + q * 4 + (e & 3);
        v[e] = (__bf16)(transposed ? w[((int64_t)c * K + k) * Cd + n] : w[((int64_t)n * K + k) * Cs + c]);
    }
    wp[idx] = v;
}

// x3 form (u3d_common.h "bf16x3"): the bf16 form's vectors, three planes per (lane, 32-channel group jj, column block nb):
// wp[((((slice*K + k)*CS32 + jj)*2 + nb)*3 + plane)*64 + lane][e] = plane `plane` of the exact split of the same W element
__device__ __forceinline__ void weight_x3_vectors(const float* __restrict__ w, bf16x8* __restrict__ wp, int64_t idx, int Cd, int K, int Cs, int transposed) {
    const int cs32 = Cs / 32;
    const int lane = (int)(idx & 63);
    int64_t t = idx >> 6;
    const int nb = (int)(t & 1); t >>= 1;
    const int jj = (int)(t % cs32); t /= cs32;
    const int k = (int)(t % K);
    const int slice = (int)(t / K);
    const int n = slice * 32 + nb * 16 + (lane & 15), q = lane >> 4;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = (2 * jj + (e >> 2)) * 16 + q * 4 + (e & 3);
        v[e] = transposed ? w[((int64_t)c * K + k) * Cd + n] : w[((int64_t)n * K + k) * Cs + c];
    }
    bf16x8 pl[3];
    split3_x8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, pl);
    const int64_t o = (idx >> 6) * 192 + lane;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) wp[o + pq * 64] = pl[pq];
}
__global__ __launch_bounds__(256) void weight_pack_x3_k(const float* __restrict__ w, bf16x8* __restrict__ wp, int Cd, int K, int Cs, int transposed) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one (lane, jj, nb) triple of vectors per thread
    if (idx >= (int64_t)(Cd / 32) * K * (Cs / 32) * 2 * 64) return;
    weight_x3_vectors(w, wp, idx, Cd, K, Cs, transposed);
}

// every convolution weight of the model, both orientations, in ONE launch (the weights change once per optimizer step; 89 single
// packs cost 0.38 ms of launch latency per step).  desc[i] = {src, dst, Cd, K, Cs, transposed, format (0 fp32, 1 bf16, 2 x3), first block}: block b belongs
// to the last descriptor whose first block is <= b.
struct PackDesc { const float* src; void* dst; int64_t Cd, K, Cs, transposed, bf, block0; };
__global__ __launch_bounds__(256) void weight_pack_batch_k(const PackDesc* __restrict__ desc, int n_desc) {
    int lo = 0, hi = n_desc;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (desc[mid].block0 <= (int64_t)blockIdx.x) lo = mid; else hi = mid;
    }
    const PackDesc d = desc[lo];
    const int64_t idx = ((int64_t)blockIdx.x - d.block0) * 256 + threadIdx.x;
    const int Cd = (int)d.Cd, K = (int)d.K, Cs = (int)d.Cs;
    const int lane = (int)(idx & 63);
    int64_t t = idx >> 6;
    const int nb = (int)(t & 1); t >>= 1;
    if (d.bf == 2) {
        if (idx >= (int64_t)(Cd / 32) * K * (Cs / 32) * 2 * 64) return;
        weight_x3_vectors(d.src, reinterpret_cast<bf16x8*>(d.dst), idx, Cd, K, Cs, (int)d.transposed);
    } else if (d.bf) {
        const int cs32 = Cs / 32;
        if (idx >= (int64_t)(Cd / 32) * K * cs32 * 2 * 64) return;
        const int jj = (int)(t % cs32); t /= cs32;
        const int k = (int)(t % K), slice = (int)(t / K);
        const int n = slice * 32 + nb * 16 + (lane & 15), q = lane >> 4;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = (2 * jj + (e >> 2)) * 16 + q * 4 + (e & 3);
            v[e] = (__bf16)(d.transposed ? d.src[((int64_t)c * K + k) * Cd + n] : d.src[((int64_t)n * K + k) * Cs + c]);
        }
        reinterpret_cast<bf16x8*>(d.dst)[idx] = v;
    } else {
        const int cs16 = Cs / 16;
        if (idx >= (int64_t)(Cd / 32) * K * cs16 * 2 * 64) return;
        const int j = (int)(t % cs16); t /= cs16;
        const int k = (int)(t % K), slice = (int)(t / K);
        const int n = slice * 32 + nb * 16 + (lane & 15), c = j * 16 + (lane >> 4) * 4;
        float4 v;
        if (!d.transposed) {
            v = *reinterpret_cast<const float4*>(d.src + ((int64_t)n * K + k) * Cs + c);
        } else {
            v.x = d.src[((int64_t)(c + 0) * K + k) * Cd + n];
            v.y = d.src[((int64_t)(c + 1) * K + k) * Cd + n];
            v.z = d.src[((int64_t)(c + 2) * K + k) * Cd + n];
            v.w = d.src[((int64_t)(c + 3) * K + k) * Cd + n];
        }
        reinterpret_cast<float4*>(d.dst)[idx] = v;
    }
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

static int spconv_plan_impl(int Cs, int Cd, int K, int64_t n_dst, int* tile_rows, int* k_groups, bool rows_kernel) {
    if (Cs % 16 || Cd % 32 || Cs <= 0 || Cd <= 0 || Cd > 256 || Cs > 256 || K <= 0 || K > 32 || n_dst <= 0 || !tile_rows || !k_groups)
        return U3D_EUNSUPPORTED;
    const int cs16 = Cs / 16;
    if (!(cs16 == 1 || cs16 == 2 || cs16 == 4 || cs16 == 6 || cs16 == 8 || cs16 == 10 || cs16 == 12 || cs16 == 16)) return U3D_EUNSUPPORTED;
    plan_gmm(Cs, Cd, K, n_dst, tile_rows, k_groups, rows_kernel);
    return U3D_OK;
}
int u3d_spconv_plan(int Cs, int Cd, int K, int64_t n_dst, int* tile_rows, int* k_groups) {
    return spconv_plan_impl(Cs, Cd, K, n_dst, tile_rows, k_groups, false);
}
int u3d_spconv_plan_bf16a(int Cs, int Cd, int K, int64_t n_dst, int* tile_rows, int* k_groups) {
    return spconv_plan_impl(Cs, Cd, K, n_dst, tile_rows, k_groups, true);
}

static int spconv_gmm_impl(const float* src, int64_t n_src, const float* w_rows, const int32_t* gather, const int32_t* scatter,
                           const int32_t* tile_starts, int K, int64_t cap, int Cs, int Cd, int64_t n_dst, int tile_rows,
                           int k_groups, const float* addend, float* dst, void* ws, float* bn_partial, double flops_hint, u3d_stream_t stream, int pr) {
    if (pr && Cs % 32) { set_error("spconv_gmm_%s: Cs=%d must be a multiple of 32", pr == 1 ? "bf16" : (pr == 2 ? "x3" : "bf16a"), Cs); return U3D_EUNSUPPORTED; }
    if (!src || !w_rows || !gather || !scatter || !tile_starts || !dst || K <= 0 || K > 32 || n_dst <= 0 || n_src <= 0 || cap <= 0) return U3D_EINVAL;
    // the kernel addresses through 32-bit buffer offsets and multiplies row indices with v_mul_u32_u24
    if (pr == 3 && bn_partial) { set_error("spconv_gmm_bf16a: per-tile statistics are not produced by this kernel"); return U3D_EUNSUPPORTED; }
    if (n_src >= (1 << 24) || n_dst >= (1 << 24) || n_src * Cs * 4 >= 0x7fffffffLL || (int64_t)K * cap * 4 >= 0x7fffffffLL) {
        set_error("spconv_gmm: %lld source rows x %d channels / %lld pairs per offset exceed the kernel's 32-bit addressing",
                  (long long)n_src, Cs, (long long)cap);
        return U3D_EUNSUPPORTED;
    }
    int R = 0, G = 0;
    if (spconv_plan_impl(Cs, Cd, K, n_dst, &R, &G, pr == 3) != U3D_OK || R != tile_rows || G != k_groups || (G > 1 && !ws)) {
        set_error("spconv_gmm: unsupported Cs=%d Cd=%d or plan mismatch (tile %d/%d groups %d/%d)", Cs, Cd, tile_rows, R, k_groups, G);
        return U3D_EUNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(U3D_K_CONV_FWD, s, flops_hint);
    GmmParams p;
    p.src = src; p.w = w_rows; p.gather = gather; p.scatter = scatter; p.ts = tile_starts; p.addend = addend;
    p.out = G > 1 ? (float*)ws : dst;
    if (bn_partial && G > 1) { set_error("spconv_gmm: per-tile statistics are only produced without offset groups (k_groups = %d)", G); return U3D_EUNSUPPORTED; }
    p.stats = bn_partial;
    p.K = K; p.cap = cap; p.Cs = Cs; p.Cd = Cd; p.n_dst = n_dst; p.n_src = n_src; p.n_sub = ceil_div(n_dst, R);
    p.n_slices = Cd / GMM_CDS; p.G = G; p.kper = (int)ceil_div(K, G);
    const int cs16 = Cs / 16;
    int rc = U3D_EUNSUPPORTED;
#define U3D_GMM_CASE(cs) if (cs16 == cs) rc = (R == 64) ? launch_gmm<cs, 64>(p, s) : launch_gmm<cs, 32>(p, s);
#define U3D_GMM_CASE_BF(cs) if (cs16 == cs) rc = (R == 64) ? launch_gmm<cs, 64, 1>(p, s) : launch_gmm<cs, 32, 1>(p, s);
#define U3D_GMM_CASE_X3(cs) if (cs16 == cs) rc = (R == 64) ? launch_gmm<cs, 64, 2>(p, s) : launch_gmm<cs, 32, 2>(p, s);
    // bf16 / three-plane operands: the workgroup-tile kernel (spconv_wg.hip, weights of an offset shared through LDS) unless the
    // launch asks for per-tile statistics (wave-tile epilogue only) or u3d_conv_kernel(0) / U3D_GMM_WG=0 selected the wave-tile kernel
    // (offset groups -- the small levels -- stay on the wave-tile kernel: there the workgroup form measured level or behind)
    // bf16 operands from fp32 rows (pr = 1): the wave-tile kernel measured ahead (level 1, 32 -> 32: 78 us against 88);
    // u3d_conv_kernel(2) takes the workgroup form wherever it is instantiated (tests, A/B runs)
    const int ck = u3d_conv_kernel(-1);
    if (pr == 3 || (pr && (ck == 2 || (ck == 1 && pr == 2 && G == 1)) && !bn_partial && gmm_wg_supported(cs16, R, pr))) {
        rc = launch_gmm_wg(p, cs16, R, pr, s);
    } else if (pr == 1) {
        U3D_GMM_CASE_BF(2) U3D_GMM_CASE_BF(4) U3D_GMM_CASE_BF(6) U3D_GMM_CASE_BF(8) U3D_GMM_CASE_BF(10) U3D_GMM_CASE_BF(12) U3D_GMM_CASE_BF(16)
    } else if (pr == 2) {
        U3D_GMM_CASE_X3(2) U3D_GMM_CASE_X3(4) U3D_GMM_CASE_X3(6) U3D_GMM_CASE_X3(8) U3D_GMM_CASE_X3(10) U3D_GMM_CASE_X3(12) U3D_GMM_CASE_X3(16)
    } else {
        U3D_GMM_CASE(1) U3D_GMM_CASE(2) U3D_GMM_CASE(4) U3D_GMM_CASE(6) U3D_GMM_CASE(8) U3D_GMM_CASE(10) U3D_GMM_CASE(12) U3D_GMM_CASE(16)
    }
#undef U3D_GMM_CASE
#undef U3D_GMM_CASE_BF
#undef U3D_GMM_CASE_X3
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

int u3d_spconv_gmm(const float* src, int64_t n_src, const float* w_rows, const int32_t* gather, const int32_t* scatter,
                   const int32_t* tile_starts, int K, int64_t cap, int Cs, int Cd, int64_t n_dst, int tile_rows,
                   int k_groups, const float* addend, float* dst, void* ws, float* bn_partial, double flops_hint, u3d_stream_t stream) {
    return spconv_gmm_impl(src, n_src, w_rows, gather, scatter, tile_starts, K, cap, Cs, Cd, n_dst, tile_rows, k_groups, addend, dst, ws,
                           bn_partial, flops_hint, stream, 0);
}

int u3d_spconv_gmm_bf16(const float* src, int64_t n_src, const void* w_rows_bf16, const int32_t* gather, const int32_t* scatter,
                        const int32_t* tile_starts, int K, int64_t cap, int Cs, int Cd, int64_t n_dst, int tile_rows,
                        int k_groups, const float* addend, float* dst, void* ws, float* bn_partial, double flops_hint, u3d_stream_t stream) {
    return spconv_gmm_impl(src, n_src, (const float*)w_rows_bf16, gather, scatter, tile_starts, K, cap, Cs, Cd, n_dst, tile_rows, k_groups,
                           addend, dst, ws, bn_partial, flops_hint, stream, 1);
}

int u3d_spconv_gmm_x3(const float* src, int64_t n_src, const void* w_rows_x3, const int32_t* gather, const int32_t* scatter,
                      const int32_t* tile_starts, int K, int64_t cap, int Cs, int Cd, int64_t n_dst, int tile_rows,
                      int k_groups, const float* addend, float* dst, void* ws, float* bn_partial, double flops_hint, u3d_stream_t stream) {
    return spconv_gmm_impl(src, n_src, (const float*)w_rows_x3, gather, scatter, tile_starts, K, cap, Cs, Cd, n_dst, tile_rows, k_groups,
                           addend, dst, ws, bn_partial, flops_hint, stream, 2);
}

int u3d_spconv_gmm_bf16a(const void* src_bf16, int64_t n_src, const void* w_rows_bf16, const int32_t* gather, const int32_t* scatter,
                         const int32_t* tile_starts, int K, int64_t cap, int Cs, int Cd, int64_t n_dst, int tile_rows,
                         int k_groups, const float* addend, float* dst, void* ws, double flops_hint, u3d_stream_t stream) {
    return spconv_gmm_impl((const float*)src_bf16, n_src, (const float*)w_rows_bf16, gather, scatter, tile_starts, K, cap, Cs, Cd, n_dst, tile_rows,
                           k_groups, addend, dst, ws, nullptr, flops_hint, stream, 3);
}

int u3d_weight_pack_x3(const float* w, void* wp, int Cd, int K, int Cs, int transposed, u3d_stream_t stream) {
    if (!w || !wp || Cd <= 0 || K <= 0 || Cs <= 0 || Cd % 32 || Cs % 32) return U3D_EINVAL;
    const int64_t total8 = (int64_t)Cd * K * Cs / 8;
    hipLaunchKernelGGL(weight_pack_x3_k, dim3((unsigned)ceil_div(total8, 256)), dim3(256), 0, (hipStream_t)stream, w, (bf16x8*)wp, Cd, K, Cs, transposed);
    return check_launch("weight_pack_x3");
}

int u3d_weight_pack_bf16(const float* w, void* wp, int Cd, int K, int Cs, int transposed, u3d_stream_t stream) {
    if (!w || !wp || Cd <= 0 || K <= 0 || Cs <= 0 || Cd % 32 || Cs % 32) return U3D_EINVAL;
    const int64_t total8 = (int64_t)Cd * K * Cs / 8;
    hipLaunchKernelGGL(weight_pack_bf16_k, dim3((unsigned)ceil_div(total8, 256)), dim3(256), 0, (hipStream_t)stream, w, (bf16x8*)wp, Cd, K, Cs, transposed);
    return check_launch("weight_pack_bf16");
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

static int spconv_wgrad_impl(const float* x, int64_t n_rows_x, const float* dy, const int32_t* rows_x, const int32_t* rows_dy,
                             const int32_t* tile_starts, int K, int64_t cap, int64_t n_rows_dy, int tile_rows, int Cs, int Cd,
                             float* dW, void* ws, double flops_hint, u3d_stream_t stream, bool bf) {
    if (!x || !dy || !rows_x || !rows_dy || !tile_starts || !dW || !ws || K <= 0 || cap <= 0 || n_rows_dy <= 0 || n_rows_x <= 0) return U3D_EINVAL;
    if (n_rows_x >= (1 << 24) || n_rows_dy >= (1 << 24) || n_rows_x * Cs * 4 >= 0x7fffffffLL || n_rows_dy * Cd * 4 >= 0x7fffffffLL ||
        (int64_t)K * cap * 4 >= 0x7fffffffLL) {
        set_error("spconv_wgrad: %lld / %lld rows exceed the kernel's 32-bit addressing", (long long)n_rows_x, (long long)n_rows_dy);
        return U3D_EUNSUPPORTED;
    }
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
    // fp32 operands: the three-plane bf16 walk only where it measured faster than the pipelined fp32 walk (tools/prof_wgrad.py,
    // 8 scenes, TFLOP/s fp32 MFMA vs x3: 32 ch 62.1 / 36.6, 64 ch 65.6 / 57.3, 96 ch 51.4 / 48.6, 128 ch 29.4 / 27.0,
    // 160 ch 9.2 / 16.0 -- the walk is bound by its two-level gathers, and splitting 8 values x (NG + NX) blocks per trip costs
    // more VALU time than the six bf16 MFMAs save below 160 channels), unless U3D_FP32_MATH=mfma
    static const int x3_min = [] { const char* e = getenv("U3D_WGRAD_X3_MIN"); return e ? atoi(e) : 160 * 160; }();
    const int pr = bf ? 1 : ((fp32_x3() && Cs * Cd >= x3_min) ? 2 : 0);
#define U3D_WG_CASE(cs, cd) if (cs16 == cs && cd16 == cd) return pr == 1 ? launch_wgrad<cs, cd, 1>(p, dW, s) : (pr == 2 ? launch_wgrad<cs, cd, 2>(p, dW, s) : launch_wgrad<cs, cd, 0>(p, dW, s));
    U3D_WG_CASE(1, 2) U3D_WG_CASE(2, 2) U3D_WG_CASE(4, 2) U3D_WG_CASE(4, 4) U3D_WG_CASE(8, 4)
    U3D_WG_CASE(6, 6) U3D_WG_CASE(12, 6) U3D_WG_CASE(8, 8) U3D_WG_CASE(16, 8) U3D_WG_CASE(10, 10)
    U3D_WG_CASE(2, 4) U3D_WG_CASE(4, 6) U3D_WG_CASE(6, 8) U3D_WG_CASE(8, 10)
    U3D_WG_CASE(6, 4) U3D_WG_CASE(8, 6) U3D_WG_CASE(10, 8)
#undef U3D_WG_CASE
    set_error("spconv_wgrad: no instantiation for Cs=%d Cd=%d", Cs, Cd);
    return U3D_EUNSUPPORTED;
}

int u3d_spconv_wgrad(const float* x, int64_t n_rows_x, const float* dy, const int32_t* rows_x, const int32_t* rows_dy,
                     const int32_t* tile_starts, int K, int64_t cap, int64_t n_rows_dy, int tile_rows, int Cs, int Cd,
                     float* dW, void* ws, double flops_hint, u3d_stream_t stream) {
    return spconv_wgrad_impl(x, n_rows_x, dy, rows_x, rows_dy, tile_starts, K, cap, n_rows_dy, tile_rows, Cs, Cd, dW, ws, flops_hint, stream, false);
}

int u3d_spconv_wgrad_bf16(const float* x, int64_t n_rows_x, const float* dy, const int32_t* rows_x, const int32_t* rows_dy,
                          const int32_t* tile_starts, int K, int64_t cap, int64_t n_rows_dy, int tile_rows, int Cs, int Cd,
                          float* dW, void* ws, double flops_hint, u3d_stream_t stream) {
    return spconv_wgrad_impl(x, n_rows_x, dy, rows_x, rows_dy, tile_starts, K, cap, n_rows_dy, tile_rows, Cs, Cd, dW, ws, flops_hint, stream, true);
}

int u3d_weight_pack_batch(const void* desc, int n_desc, int64_t total_blocks, u3d_stream_t stream) {
    if (!desc || n_desc <= 0 || total_blocks <= 0 || total_blocks >= 0x7fffffffLL) return U3D_EINVAL;
    hipLaunchKernelGGL(weight_pack_batch_k, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const PackDesc*)desc, n_desc);
    return check_launch("weight_pack_batch");
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
