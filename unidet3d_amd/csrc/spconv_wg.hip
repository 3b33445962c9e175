// Sparse convolution forward / input gradient, workgroup-tile form (round 4): the same output-stationary gather-MFMA-scatter
// program as spconv.hip's spconv_gmm_k, but a WORKGROUP (four waves) owns 4R consecutive dst rows x one 32-column slice and walks
// the kernel offsets in lock step, so that
//   * the packed weight fragments of an offset are fetched ONCE per workgroup into LDS and read from there by the four waves.
//     In spconv_gmm_k every <= 32-pair item re-reads its 6 KB (three bf16 planes) of fragments through the vector-memory path --
//     48 of the ~86 cache lines an item moves through the CU's texture path, which is what paces that kernel (~4 cycles per
//     128-byte line and CU: 311 cycles per item and CU measured, 344 by that count).  Here the fragments cost
//     six ds_read_b128 per unit (256 B/clk) and the workgroup's one copy of them 48 lines per OFFSET;
//   * the pairs of an offset inside the workgroup's rows are dealt to the four waves in equal shares of 16-pair chunks,
//     whichever 64-row band they scatter to: the accumulator tile [4R][32] fp32 is shared by the workgroup.  Inside one offset
//     every dst row occurs at most once, so the waves' read-modify-writes never meet; between offsets there is a barrier -- the
//     one the weight hand-over needs anyway.  Shares differ by at most one chunk (wave-private tiles: 0 .. 4 chunks per offset).
// The item / unit pipeline of a wave (indices two items ahead, rows one unit ahead, transposition through a private LDS image,
// accumulation through the MFMA C operand) is spconv_gmm_k's; see the comments there.  Replaces spconv's implicit-GEMM kernels
// behind unidet3d/spconv_unet.py:34-72,146-192 (reference), for bf16 operands and for fp32 products from three bf16 planes.
#include <stdlib.h>

#include "spconv_gmm.h"

namespace u3d {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ __forceinline__ f32x4 wg_mfma(const f32x4& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), b, c, 0, 0, 0);
}

// float index of (row r, 16-byte quad c4) in the accumulator tile: 128-byte rows, quad stored at c4 ^ (r & 7)
__device__ __forceinline__ int wg_acc_idx(int r, int c4) { return r * 32 + ((c4 ^ (r & 7)) << 2); }

__device__ __forceinline__ void wg_barrier() {      // LDS traffic of this wave done, then the workgroup barrier; vmcnt is NOT drained
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// LDS of a workgroup: accumulator tile (4R rows + one scratch row per wave) | 4 staging images (16 rows x 128 B) | weight slot(s)
constexpr int wg_acc_floats(int r) { return (4 * r + 4) * 32; }
constexpr int wg_stage_floats(int pr) { return pr == 3 ? 0 : 16 * 8 * 4; }      // bf16 rows arrive as MFMA fragments: no staging image
constexpr int wg_slot_bytes(int cs16, int pr) { return (cs16 / 2) * (pr == 2 ? 6144 : 2048); }
constexpr int wg_fixed_bytes(int r, int pr) { return (wg_acc_floats(r) + 4 * wg_stage_floats(pr)) * 4; }
constexpr int wg_per_cu(int bytes) { return 160 * 1024 / bytes > 8 ? 8 : 160 * 1024 / bytes; }      // workgroups per CU by LDS (8 x 4 waves: the CU's wave slots)
// two weight slots (one barrier per offset) unless the second slot costs a workgroup per CU (then one slot, two barriers)
constexpr int wg_nslot(int cs16, int r, int pr) {
    return wg_per_cu(wg_fixed_bytes(r, pr) + 2 * wg_slot_bytes(cs16, pr)) >= wg_per_cu(wg_fixed_bytes(r, pr) + wg_slot_bytes(cs16, pr)) ? 2 : 1;
}
constexpr int wg_lds_bytes(int cs16, int r, int pr) {
    return wg_fixed_bytes(r, pr) + wg_nslot(cs16, r, pr) * wg_slot_bytes(cs16, pr);
}

struct WgItem {
    int k, base, e;      // wave-uniform; when !valid the fields still name a real item (addresses stay legal)
    bool valid;
};

// PR = 1: bf16 operands (rows rounded as the operand is formed, weights pre-rounded); PR = 2: three exact bf16 planes per operand.
// PR = 3: the source rows are ALREADY bf16 in HBM (the shadow copy a batch-norm kernel wrote next to its fp32 output, [n][Cs] bf16):
// lane (pair i16, lane group q) loads its MFMA fragment -- 16 bytes at byte 16 q of the unit's 64, which the shadow's fragment
// order (bn.hip store_shadow) fills with channels 4q .. 4q+3 and 16+4q .. 16+4q+3 -- straight from the row: no staging image, no LDS
// transposition, no rounding arithmetic, half the gathered bytes.  Same operands in the same k slots as PR = 1, the same packed
// weights (u3d_weight_pack_bf16): bit-identical results.
// A unit is one 32-channel group of the source channels (JB = 2 in spconv_gmm_k's terms).
template <int CS16, int R, int PR>
struct GmmWgWave {
    static constexpr bool X3 = PR == 2, BR = PR == 3;
    static constexpr int NCH = R / 32;            // 16-pair chunks per item
    static constexpr int NJB = CS16 / 2;          // units per item
    static constexpr int W = 16 * NCH;            // pairs per item
    static constexpr int NI = BR ? NCH : W / 8;   // load instructions per unit: 8 rows x 128 B each (fp32 rows) / one fragment per chunk (bf16 rows)
    static constexpr int TR = 4 * R;
    static constexpr int NPL = X3 ? 3 : 1;        // planes per weight fragment
    static constexpr int WB = NPL * 2048;         // bytes of packed weights per (offset, 32-channel group): 2 column blocks x NPL x 1 KB
    static constexpr int WSLOT = NJB * WB;
    static constexpr int NSLOT = wg_nslot(CS16, R, PR);
    static constexpr int NPIECE = WSLOT / 16, NP = (NPIECE + 255) / 256;      // 16-byte pieces of a slot; per thread
    static constexpr bool PREF = NP <= 6;         // next offset's weights wait in registers (otherwise fetched between the barriers)

    struct Buf { f32x4 a[NI]; };                  // instruction i: rows i*8 + lane/8 of the item, 16-byte piece lane%8 of the unit

    __amdgpu_buffer_rsrc_t rs_src, rs_g, rs_s, rs_w;
    char* accq;                                   // accumulator tile
    float* stage;                                 // this wave's 16 rows x 8 pieces, piece slot p ^ (row & 7)
    char* wslot;                                  // first weight slot
    int lane, i16, tid, wave, K, row0, cs4, q16;
    int64_t cap;
    int k_hi;
    int cs2;                                      // bf16 rows: bytes per source row
    int ts_s, ts_e;                               // lane k: THIS WAVE's share of the pairs of offset k in the workgroup's rows
    unsigned kmask;                               // offsets with pairs in the workgroup's rows (= the steps; the same in all four waves)
    int kc, step;                                 // current step's offset (32: past the last), its ordinal
    int w_soff0;                                  // byte offset of (slice, offset 0) in the packed weights
    int cg, cs_;                                  // byte offsets of this lane's first entries in an item's index window
    int wr_off, rd_off[2];                        // float offsets into `stage` of this lane's write / fragment reads
    int wrd;                                      // byte offset (from wslot) of this lane's fragment reads in the current slot
    f32x4 wreg[PREF ? NP : 1];                    // this thread's pieces of the NEXT step's weights

    WgItem it0, it1;
    int g_cur[NI];
    int ix1_g[NI], ix1_s[NCH];
    int soff0, soff1;
    f32x4 d00, d01, d10, d11;

    __device__ __forceinline__ bool range_of(int k, int& s_, int& e_) const {
        s_ = __builtin_amdgcn_readlane(ts_s, k);
        e_ = __builtin_amdgcn_readlane(ts_e, k);
        return s_ < e_;
    }
    __device__ __forceinline__ WgItem first_item(int k_lo) const {
        WgItem n{k_lo, 0, 0, false};
        for (int k = k_lo; k < k_hi; ++k)
            if (range_of(k, n.base, n.e)) { n.k = k; n.valid = true; break; }
        return n;
    }
    __device__ __forceinline__ WgItem next_of(const WgItem& it) const {
        WgItem n = it;
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
    __device__ __forceinline__ void load_idx(const WgItem& it, int (&g)[NI], int (&s_)[NCH]) const {
        const int soff_k = (int)(it.k * cap) * 4;
        const int vg = cg + it.base * 4, vs = cs_ + it.base * 4;
        if constexpr (BR) {          // the lane's own pair of every chunk, gather and scatter side alike
#pragma unroll
            for (int c = 0; c < NCH; ++c) g[c] = bload32(rs_g, vs + c * 64, soff_k);
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) g[i] = bload32(rs_g, vg + i * 32, soff_k);
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) s_[c] = bload32(rs_s, vs + c * 64, soff_k);
    }
    __device__ __forceinline__ void issue(Buf& buf, const int (&g)[NI], int u) const {
        if constexpr (BR) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) buf.a[c] = bload128(rs_src, (int)__umul24(g[c], cs2) + q16, u * 64);
            return;
        }
        const int lp16 = (lane & 7) * 16;
#pragma unroll
        for (int i = 0; i < NI; ++i) buf.a[i] = bload128(rs_src, (int)__umul24(g[i], cs4) + lp16, u * 128);
    }
    static __device__ __forceinline__ bf16x8 cvt8(const f32x4& lo, const f32x4& hi) {
        return bf16x8{(__bf16)lo[0], (__bf16)lo[1], (__bf16)lo[2], (__bf16)lo[3], (__bf16)hi[0], (__bf16)hi[1], (__bf16)hi[2], (__bf16)hi[3]};
    }
    struct Frag { f32x4 v[2]; };
    // rows of one 16-pair chunk: registers -> swizzled LDS image -> fragments (in-order LDS queue of one wave: no barrier)
    template <int C>
    __device__ __forceinline__ Frag frags(const Buf& buf) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(stage + wr_off + i * 256) = buf.a[C * 2 + i];
        Frag f;
#pragma unroll
        for (int j = 0; j < 2; ++j) f.v[j] = *reinterpret_cast<const f32x4*>(stage + rd_off[j]);
        return f;
    }
    // byte offsets of this lane's accumulator rows (chunk 0 / 1) of item `it`; lanes past the end of the wave's share use its scratch row
    __device__ __forceinline__ void row_offsets(const WgItem& it, const int (&s_)[NCH], int& o0, int& o1) const {
        const int left = it.e - it.base;
        const int r0 = i16 < left ? s_[0] - row0 : TR + wave;
        const int r1 = NCH == 2 ? (16 + i16 < left ? s_[NCH - 1] - row0 : TR + wave) : 0;
        o0 = ((r0 << 7) | ((r0 & 7) << 4)) ^ q16;
        o1 = NCH == 2 ? ((r1 << 7) | ((r1 & 7) << 4)) ^ q16 : 0;
    }

    // ---- weights of a step: global -> registers -> LDS slot ----
    __device__ __forceinline__ void load_w(int k, f32x4 (&r)[PREF ? NP : 1]) const {
        if constexpr (PREF) {
            const int so = w_soff0 + k * WSLOT;
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if ((i + 1) * 256 <= NPIECE || tid + i * 256 < NPIECE) r[i] = bload128(rs_w, tid * 16 + i * 4096, so);
        }
    }
    __device__ __forceinline__ void write_w(int slot, const f32x4 (&r)[PREF ? NP : 1]) const {
        if constexpr (PREF) {
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if ((i + 1) * 256 <= NPIECE || tid + i * 256 < NPIECE) *reinterpret_cast<f32x4*>(wslot + slot * WSLOT + tid * 16 + i * 4096) = r[i];
        }
    }
    __device__ __forceinline__ void copy_w(int k, int slot) const {      // large slots: straight through, a few pieces at a time
        const int so = w_soff0 + k * WSLOT;
#pragma unroll 1
        for (int i0 = 0; i0 < NP; i0 += 4) {
            f32x4 t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i0 + i < NP && tid + (i0 + i) * 256 < NPIECE) t[i] = bload128(rs_w, tid * 16 + (i0 + i) * 4096, so);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i0 + i < NP && tid + (i0 + i) * 256 < NPIECE) *reinterpret_cast<f32x4*>(wslot + slot * WSLOT + tid * 16 + (i0 + i) * 4096) = t[i];
        }
    }
    static __device__ __forceinline__ int next_k(unsigned mask, int after) {      // lowest set bit above `after` (32: none)
        const unsigned rest = after >= 31 ? 0u : mask & (~1u << after);
        return rest ? __builtin_ctz(rest) : 32;
    }
    // Before the first step: weights of the first offset into slot 0, the second offset's on their way; the barrier also
    // publishes the accumulator initialisation.
    __device__ __forceinline__ void first_step() {
        kc = kmask ? __builtin_ctz(kmask) : 32;
        step = 0;
        if (kc < 32) {
            if constexpr (PREF) {
                load_w(kc, wreg);
                write_w(0, wreg);
                const int kn = next_k(kmask, kc);
                if (kn < 32) load_w(kn, wreg);
            } else {
                copy_w(kc, 0);
            }
        }
        wrd = lane * 16;
        wg_barrier();
    }
    // Leave step kc: every wave calls this once per step, in the same order (the barrier count is the step count).
    __device__ __forceinline__ void advance() {
        const int kn = next_k(kmask, kc);
        const int slot = NSLOT == 2 ? ((step + 1) & 1) : 0;
        if constexpr (NSLOT == 1) wg_barrier();             // every wave has read the slot for the last time (and written its rows)
        // two slots: the other slot was last read in step - 1, and every wave has passed the barrier that ended that step
        if (kn < 32) {
            if constexpr (PREF) {
                write_w(slot, wreg);
                const int kn2 = next_k(kmask, kn);
                if (kn2 < 32) load_w(kn2, wreg);
            } else {
                copy_w(kn, slot);
            }
        }
        wg_barrier();                                       // accumulator rows of step kc and the next weights are visible
        kc = kn;
        ++step;
        wrd = slot * WSLOT + lane * 16;
    }
    __device__ __forceinline__ void sync_to(int k) {
        while (kc < k && kc < 32) advance();
    }

    template <int U>
    __device__ __forceinline__ void unit(Buf& cur, Buf& nxt) {
        const bool two = NCH == 2 && it0.base + 16 < it0.e;
        if constexpr (U == 0) {                // accumulator rows of the item -> C operands
            d00 = *reinterpret_cast<const f32x4*>(accq + soff0);
            d01 = *reinterpret_cast<const f32x4*>(accq + (soff0 ^ 64));
            if (two) {
                d10 = *reinterpret_cast<const f32x4*>(accq + soff1);
                d11 = *reinterpret_cast<const f32x4*>(accq + (soff1 ^ 64));
            }
        }
        Frag f0, f1;
        if constexpr (!BR) {
            f0 = frags<0>(cur);
            f1 = f0;
            if (two) f1 = frags<NCH - 1>(cur);
        }
        f32x4 wf[2][NPL];                      // [column block][plane] of this unit's 32-channel group, from the workgroup's slot
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int q = 0; q < NPL; ++q) wf[nb][q] = *reinterpret_cast<const f32x4*>(wslot + wrd + U * WB + (nb * NPL + q) * 1024);
        WgItem it2;
        int g2[NI], s2[NCH], n0 = 0, n1 = 0;
        if constexpr (U == NJB - 1) {
            row_offsets(it1, ix1_s, n0, n1);
            it2 = next_of(it1);
            load_idx(it2, g2, s2);
            issue(nxt, ix1_g, 0);
        } else {
            issue(nxt, g_cur, U + 1);
        }
        if constexpr (X3) {
            // h.h goes to the running row (the MFMA's C operand), the five low-order plane products to a zero-initialised
            // accumulator added at the end of the unit (spconv.hip / DESIGN.md 4.11: the bf16 MFMA truncates products against C)
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            f32x4 t00 = z, t01 = z, t10 = z, t11 = z;
            bf16x8 x0[3];
            split3_x8(f0.v[0], f0.v[1], x0);
#pragma unroll
            for (int o = 2; o >= 1; --o)
#pragma unroll
                for (int qa = 0; qa <= o; ++qa) {
                    t00 = wg_mfma(wf[0][qa], x0[o - qa], t00);
                    t01 = wg_mfma(wf[1][qa], x0[o - qa], t01);
                }
            d00 = wg_mfma(wf[0][0], x0[0], d00);
            d01 = wg_mfma(wf[1][0], x0[0], d01);
            if (two) {
                bf16x8 x1[3];
                split3_x8(f1.v[0], f1.v[1], x1);
#pragma unroll
                for (int o = 2; o >= 1; --o)
#pragma unroll
                    for (int qa = 0; qa <= o; ++qa) {
                        t10 = wg_mfma(wf[0][qa], x1[o - qa], t10);
                        t11 = wg_mfma(wf[1][qa], x1[o - qa], t11);
                    }
                d10 = wg_mfma(wf[0][0], x1[0], d10);
                d11 = wg_mfma(wf[1][0], x1[0], d11);
                d10 += t10; d11 += t11;
            }
            d00 += t00; d01 += t01;
        } else {
            bf16x8 x0, x1;
            if constexpr (BR) {
                x0 = __builtin_bit_cast(bf16x8, cur.a[0]);
                x1 = __builtin_bit_cast(bf16x8, cur.a[NCH - 1]);
            } else {
                x0 = cvt8(f0.v[0], f0.v[1]);
                x1 = cvt8(f1.v[0], f1.v[1]);
            }
            d00 = wg_mfma(wf[0][0], x0, d00);
            d01 = wg_mfma(wf[1][0], x0, d01);
            if (two) {
                d10 = wg_mfma(wf[0][0], x1, d10);
                d11 = wg_mfma(wf[1][0], x1, d11);
            }
        }
        if constexpr (U == NJB - 1) {
            *reinterpret_cast<f32x4*>(accq + soff0) = d00;
            *reinterpret_cast<f32x4*>(accq + (soff0 ^ 64)) = d01;
            if (two) {
                *reinterpret_cast<f32x4*>(accq + soff1) = d10;
                *reinterpret_cast<f32x4*>(accq + (soff1 ^ 64)) = d11;
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) { g_cur[i] = ix1_g[i]; ix1_g[i] = g2[i]; }
#pragma unroll
            for (int c = 0; c < NCH; ++c) ix1_s[c] = s2[c];
            soff0 = n0; soff1 = n1;
            it0 = it1; it1 = it2;
        }
    }
    template <int U>
    __device__ __forceinline__ void units(Buf& a, Buf& b) {
        unit<U>(a, b);
        if constexpr (U + 1 < NJB) units<U + 1>(b, a);
    }

    __device__ __forceinline__ void run(int k_lo) {
        first_step();
        it0 = first_item(k_lo);
        if (it0.valid) {
            int s_first[NCH];
            load_idx(it0, g_cur, s_first);
            it1 = next_of(it0);
            load_idx(it1, ix1_g, ix1_s);
            row_offsets(it0, s_first, soff0, soff1);
            Buf X, Y;
            issue(X, g_cur, 0);
            while (true) {
                sync_to(it0.k);                // the item's offset is a step: the loop stops exactly there
                units<0>(X, Y);
                if (!it0.valid) break;
                if constexpr (NJB % 2 == 1) {  // an odd unit count leaves the next item's first unit in Y
                    sync_to(it0.k);
                    units<0>(Y, X);
                    if (!it0.valid) break;
                }
            }
        }
        sync_to(32);                           // remaining steps (other waves' pairs) and the barrier in front of the write-out
    }
};

template <int CS16, int R, int PR>
__global__ __launch_bounds__(256) void spconv_gmm_wg_k(GmmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using WV = GmmWgWave<CS16, R, PR>;
    constexpr int TR = 4 * R;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* acc = smem;

    const int64_t wid = xcd_swizzle(blockIdx.x, gridDim.x);            // neighbouring row tiles (and the slices of one) share an XCD / L2
    const int per_sub = p.n_slices * p.G;
    const int64_t sub = wid / per_sub;                                  // tile of 4R rows
    const int rem = (int)(wid % per_sub);
    const int slice = rem / p.G, g = rem % p.G;
    const int n0 = slice * GMM_CDS;
    const int64_t row0 = sub * TR;
    const int rows = (int)min((int64_t)TR, p.n_dst - row0);
    const int64_t tsld = p.n_sub + 1;                                   // p.n_sub: R-row tiles (the tile_starts granularity)
    const int64_t t0 = sub * 4, t1 = min(sub * 4 + 4, p.n_sub);

    // ---- accumulator init: zeros, or the fused residual addend (single offset group only) ----
    for (int idx = tid; idx < rows * (GMM_CDS / 4); idx += 256) {
        const int r = idx >> 3, c4 = idx & 7;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.addend && p.G == 1) v = *reinterpret_cast<const float4*>(p.addend + (row0 + r) * p.Cd + n0 + c4 * 4);
        *reinterpret_cast<float4*>(acc + wg_acc_idx(r, c4)) = v;
    }

    WV w;
    w.rs_src = make_rsrc(p.src, p.n_src * p.Cs * (PR == 3 ? 2 : 4)); w.rs_g = make_rsrc(p.gather, (int64_t)p.K * p.cap * 4);
    w.rs_s = make_rsrc(p.scatter, (int64_t)p.K * p.cap * 4); w.rs_w = make_rsrc(p.w);
    w.accq = reinterpret_cast<char*>(acc);
    w.stage = smem + wg_acc_floats(R) + wave * wg_stage_floats(PR);
    w.wslot = reinterpret_cast<char*>(smem + wg_acc_floats(R) + 4 * wg_stage_floats(PR));
    w.lane = lane; w.i16 = lane & 15; w.tid = tid; w.wave = wave; w.K = p.K; w.cap = p.cap; w.row0 = (int)row0; w.cs4 = p.Cs * 4; w.cs2 = p.Cs * 2;
    const int q = lane >> 4, lr = lane >> 3;
    w.q16 = q << 4;
    w.cg = lr * 4; w.cs_ = (lane & 15) * 4;
    w.wr_off = (lr * 8 + ((lane & 7) ^ (lr & 7))) * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) w.rd_off[j] = ((lane & 15) * 8 + ((j * 4 + q) ^ (lane & 7))) * 4;
    w.w_soff0 = slice * p.K * WV::WSLOT;
    const int k_lo = g * p.kper;
    w.k_hi = min(p.K, k_lo + p.kper);
    // lane k: the pairs of offset k inside the workgroup's rows, and this wave's share of their 16-pair chunks
    int s = 0, e = 0;
    if (lane >= k_lo && lane < w.k_hi) {
        s = p.ts[lane * tsld + t0];
        e = p.ts[lane * tsld + t1];
    }
    w.kmask = (unsigned)__ballot(e > s);
    const int n = (e - s + 15) >> 4;
    const int c0 = (wave * n) >> 2, c1 = ((wave + 1) * n) >> 2;
    w.ts_s = s + 16 * c0;
    w.ts_e = min(e, s + 16 * c1);
    w.run(k_lo);

    float* out = p.out + (p.G > 1 ? (int64_t)g * p.n_dst * p.Cd : 0);
    for (int idx = tid; idx < rows * (GMM_CDS / 4); idx += 256) {
        const int r = idx >> 3, c4 = idx & 7;
        *reinterpret_cast<float4*>(out + (row0 + r) * p.Cd + n0 + c4 * 4) = *reinterpret_cast<const float4*>(acc + wg_acc_idx(r, c4));
    }
}

template <int CS16, int R, int PR>
static int launch_wg(const GmmParams& p, hipStream_t s) {
    constexpr int lds = wg_lds_bytes(CS16, R, PR);
    static_assert(lds <= 160 * 1024, "workgroup tile exceeds the LDS");
    static DeviceOnce attr_set;              // the attribute is per DEVICE
    int dev = 0;
    hipGetDevice(&dev);
    if (attr_set.needed(dev)) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&spconv_gmm_wg_k<CS16, R, PR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set.done(dev);
    }
    const int64_t wgs = ceil_div(p.n_sub, 4) * p.n_slices * p.G;
    hipLaunchKernelGGL((spconv_gmm_wg_k<CS16, R, PR>), dim3((unsigned)wgs), dim3(256), lds, s, p);
    return check_launch("spconv_gmm_wg");
}

bool gmm_wg_supported(int cs16, int R, int pr) {
    if (pr != 1 && pr != 2 && pr != 3) return false;
    if (R != 32 && R != 64) return false;
    return cs16 == 2 || cs16 == 4 || cs16 == 6 || cs16 == 8 || cs16 == 10 || cs16 == 12 || cs16 == 16;
}

int launch_gmm_wg(const GmmParams& p, int cs16, int R, int pr, hipStream_t s) {
#define U3D_WG_CASE(cs) \
    if (cs16 == cs) { \
        if (pr == 2) return R == 64 ? launch_wg<cs, 64, 2>(p, s) : launch_wg<cs, 32, 2>(p, s); \
        if (pr == 3) return R == 64 ? launch_wg<cs, 64, 3>(p, s) : launch_wg<cs, 32, 3>(p, s); \
        return R == 64 ? launch_wg<cs, 64, 1>(p, s) : launch_wg<cs, 32, 1>(p, s); \
    }
    U3D_WG_CASE(2) U3D_WG_CASE(4) U3D_WG_CASE(6) U3D_WG_CASE(8) U3D_WG_CASE(10) U3D_WG_CASE(12) U3D_WG_CASE(16)
#undef U3D_WG_CASE
    set_error("spconv_gmm_wg: no instantiation for Cs=%d", cs16 * 16);
    return U3D_EUNSUPPORTED;
}

}  // namespace u3d
