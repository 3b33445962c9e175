// R12 on the device in five launches: matcher + classification / box losses of all decoder layers and scenes of a
// single-dataset batch, forward AND gradients (the reference: unidet3d/criterion.py:44-178 UniDet3DCriterion.get_layer_loss /
// __call__, :200-320 the cost classes and UniMatcher, unidet3d/axis_aligned_iou_loss.py:14-53 axis_aligned_diou_loss; yaw-free
// boxes).  The torch-op formulation of the same arithmetic (unidet3d_amd/criterion.py::_loss_packed) costs ~380 launches of
// [layers, scenes, n, g] tensor ops per step; here
//   crit_cost_k   one thread per (layer, query): log-sum-exp of the logits, cost[q][j] of every GT of the query's scene
//                 (-softmax[label_j] w_cls + (1 - IoU + centre term of GT 0 -- the reference's `[:, 0]` quirk) w_box, 1e8 where the
//                 GT's query mask forbids the query);
//   crit_kth_k    one wave per (layer, scene, GT): the (topk+1)-th smallest cost of the column by k rounds of lexicographic
//                 (value, index) minimum extraction -- the threshold of `cost < values` (:316-319);
//   crit_stats_k  one workgroup per (layer, scene): matched set (a 64-bit GT mask per query), class target = label of the LAST
//                 matched GT (:98-99 index assignment), weighted cross-entropy sums and the DIoU sum over matched pairs, reduced in a
//                 fixed order (no atomics);
//   crit_final_k  one thread: the scalar loss (means over scenes / scenes with matches, sum over layers) and the per-layer scale
//                 factors of the gradients;
//   crit_grad_k   one thread per (layer, query): d loss / d logits (softmax - one-hot, weighted) and d loss / d box (analytic DIoU
//                 derivative through _bbox_to_loss, sub-gradients of min / max / clamp as torch takes them).
#include <math.h>

#include "u3d_common.h"

namespace u3d {

struct CritParams {
    const float* cls;          // [L][n_tot][C1]
    const float* box;          // [L][n_tot][6]  (centre, size)
    const int32_t* cu;         // [B+1] first query of every scene
    const int32_t* gt_off;     // [B+1] first GT of every scene
    const int64_t* gt_labels;  // [G]
    const float* gt_boxes;     // [G][6]
    const uint8_t* qmask;      // scene b: [g_b][n_b] at qm_off[b]
    const int64_t* qm_off;     // [B+1]; also the offset of scene b's [n_b][g_b] block in `cost`
    int L, B, C1, topk;
    int64_t n_tot, G, P;       // P = sum n_b g_b
    float w_cls, w_box, non_obj_w, ds_w, lw_cls, lw_box;
    // workspace
    float* cost;               // [L][P]
    float* logz;               // [L][n_tot]
    float* kth;                // [L][G]
    unsigned long long* mm;    // [L][n_tot] matched-GT bit mask
    float* stats;              // [L][B][4]: sum w, sum w nll, matched pairs, sum diou
    float* scale;              // [L][2]: (unused, box scale) -- written by crit_final_k
    float* loss;               // [1]
    float* dcls;
    float* dbox;
};

__device__ __forceinline__ int scene_of(const int32_t* cu, int B, int q) {
    int lo = 0, hi = B;                    // cu[lo] <= q < cu[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cu[mid] <= q) lo = mid; else hi = mid;
    }
    return lo;
}

struct Box6 { float p1[3], p2[3]; };
__device__ __forceinline__ Box6 corners(const float* b) {      // _bbox_to_loss (criterion.py:180-198)
    Box6 r;
#pragma unroll
    for (int a = 0; a < 3; ++a) { r.p1[a] = b[a] - b[3 + a] / 2; r.p2[a] = b[a] + b[3 + a] / 2; }
    return r;
}
__device__ __forceinline__ float iou3(const Box6& p, const Box6& t) {      // mmdet3d AxisAlignedBboxOverlaps3D(is_aligned=True), eps 1e-6
    float inter = 1.f, vp = 1.f, vt = 1.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        inter *= fmaxf(fminf(p.p2[a], t.p2[a]) - fmaxf(p.p1[a], t.p1[a]), 0.f);
        vp *= p.p2[a] - p.p1[a];
        vt *= t.p2[a] - t.p1[a];
    }
    return inter / fmaxf(vp + vt - inter, 1e-6f);
}
__device__ __forceinline__ float centre_term(const Box6& p, const Box6& t) {     // r2 / c2 (axis_aligned_iou_loss.py:30-49)
    float r2 = 0.f, c2 = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float d = (p.p1[a] + p.p2[a]) / 2 - (t.p1[a] + t.p2[a]) / 2;
        const float e = fminf(p.p1[a], t.p1[a]) - fmaxf(p.p2[a], t.p2[a]);
        r2 += d * d;
        c2 += e * e;
    }
    return r2 / c2;
}

__global__ __launch_bounds__(256) void crit_cost_k(CritParams p) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)p.L * p.n_tot) return;
    const int l = (int)(idx / p.n_tot), q = (int)(idx % p.n_tot);
    const float* x = p.cls + idx * p.C1;
    float mx = -INFINITY;
    for (int c = 0; c < p.C1; ++c) mx = fmaxf(mx, x[c]);
    float se = 0.f;
    for (int c = 0; c < p.C1; ++c) se += expf(x[c] - mx);
    p.logz[idx] = mx + logf(se);
    const int b = scene_of(p.cu, p.B, q);
    const int g0 = p.gt_off[b], g = p.gt_off[b + 1] - g0;
    if (g == 0) return;
    const int ql = q - p.cu[b], n = p.cu[b + 1] - p.cu[b];
    const Box6 pb = corners(p.box + idx * 6);
    const float rc0 = centre_term(pb, corners(p.gt_boxes + (int64_t)g0 * 6));       // the `[:, 0]` term: GT 0 of the scene for every GT
    float* crow = p.cost + (int64_t)l * p.P + p.qm_off[b] + (int64_t)ql * g;
    const uint8_t* qm = p.qmask + p.qm_off[b];
    for (int j = 0; j < g; ++j) {
        float c = 1e8f;
        if (qm[(int64_t)j * n + ql]) {
            const float prob = expf(x[p.gt_labels[g0 + j]] - mx) / se;
            const float iou_loss = 1.f - iou3(pb, corners(p.gt_boxes + (int64_t)(g0 + j) * 6));
            c = -prob * p.w_cls + (iou_loss + rc0) * p.w_box;
        }
        crow[j] = c;
    }
}

// one wave per (layer, GT): kth = (topk+1)-th smallest cost of the GT's column
__global__ __launch_bounds__(64) void crit_kth_k(CritParams p) {
    const int64_t wid = blockIdx.x;
    if (wid >= (int64_t)p.L * p.G) return;
    const int l = (int)(wid / p.G), gj = (int)(wid % p.G);
    int b = 0;
    while (b + 1 < p.B && p.gt_off[b + 1] <= gj) ++b;
    const int g0 = p.gt_off[b], g = p.gt_off[b + 1] - g0, j = gj - g0;
    const int n = p.cu[b + 1] - p.cu[b];
    const float* col = p.cost + (int64_t)l * p.P + p.qm_off[b] + j;
    const int lane = threadIdx.x;
    float pv = -INFINITY;
    int pi = -1;
    for (int r = 0; r <= p.topk; ++r) {            // r-th smallest in (value, index) order, duplicates counted
        float bv = INFINITY;
        int bi = 0x7fffffff;
        for (int q = lane; q < n; q += 64) {
            const float v = col[(int64_t)q * g];
            const bool after = v > pv || (v == pv && q > pi);
            if (after && (v < bv || (v == bv && q < bi))) { bv = v; bi = q; }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        pv = bv; pi = bi;
    }
    if (lane == 0) p.kth[(int64_t)l * p.G + gj] = pv;
}

__global__ __launch_bounds__(256) void crit_stats_k(CritParams p) {
    const int l = blockIdx.x / p.B, b = blockIdx.x % p.B;
    const int q0 = p.cu[b], n = p.cu[b + 1] - q0;
    const int g0 = p.gt_off[b], g = p.gt_off[b + 1] - g0;
    const int n_cls = p.C1 - 1;
    float sw = 0.f, swn = 0.f, cnt = 0.f, sd = 0.f;
    for (int ql = threadIdx.x; ql < n; ql += 256) {
        const int64_t idx = (int64_t)l * p.n_tot + q0 + ql;
        unsigned long long m = 0ull;
        int last = -1;
        if (g) {
            const float* crow = p.cost + (int64_t)l * p.P + p.qm_off[b] + (int64_t)ql * g;
            const float* kth = p.kth + (int64_t)l * p.G + g0;
            const Box6 pb = corners(p.box + idx * 6);
            for (int j = 0; j < g; ++j)
                if (crow[j] < kth[j]) {
                    m |= 1ull << j;
                    last = j;
                    const Box6 tb = corners(p.gt_boxes + (int64_t)(g0 + j) * 6);
                    sd += (1.f - iou3(pb, tb)) + centre_term(pb, tb);
                    cnt += 1.f;
                }
        }
        p.mm[idx] = m;
        const int target = last >= 0 ? (int)p.gt_labels[g0 + last] : n_cls;
        const float w = target == n_cls ? p.non_obj_w : 1.f;
        sw += w;
        swn += w * (p.logz[idx] - p.cls[idx * p.C1 + target]);
    }
    // fixed-order block reduction
    __shared__ float red[4][256];
    red[0][threadIdx.x] = sw; red[1][threadIdx.x] = swn; red[2][threadIdx.x] = cnt; red[3][threadIdx.x] = sd;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (threadIdx.x < s)
#pragma unroll
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 4) p.stats[((int64_t)l * p.B + b) * 4 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void crit_final_k(CritParams p) {
    if (threadIdx.x || blockIdx.x) return;
    float total = 0.f;
    for (int l = 0; l < p.L; ++l) {
        float cls_sum = 0.f, box_sum = 0.f, n_has = 0.f;
        for (int b = 0; b < p.B; ++b) {
            const float* s = p.stats + ((int64_t)l * p.B + b) * 4;
            cls_sum += p.ds_w * (s[1] / s[0]);
            if (s[2] > 0.f) { box_sum += p.ds_w * (s[3] / s[2]); n_has += 1.f; }
        }
        total += p.lw_cls * (cls_sum / p.B) + p.lw_box * (box_sum / fmaxf(n_has, 1.f));
        p.scale[l * 2 + 1] = n_has;
    }
    p.loss[0] = total;
}

__device__ __forceinline__ float pick_gt(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }     // d max(a, b) / d a
__device__ __forceinline__ float pick_lt(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }     // d min(a, b) / d a

__global__ __launch_bounds__(256) void crit_grad_k(CritParams p) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)p.L * p.n_tot) return;
    const int l = (int)(idx / p.n_tot), q = (int)(idx % p.n_tot);
    const int b = scene_of(p.cu, p.B, q);
    const int g0 = p.gt_off[b];
    const int n_cls = p.C1 - 1;
    const float* st = p.stats + ((int64_t)l * p.B + b) * 4;
    const unsigned long long m = p.mm[idx];
    // ---- classification: lw_cls * ds_w / B * w_q / sum_w * (softmax - onehot(target)) ----
    const int last = m ? 63 - __clzll(m) : -1;
    const int target = last >= 0 ? (int)p.gt_labels[g0 + last] : n_cls;
    const float w = target == n_cls ? p.non_obj_w : 1.f;
    const float kc = p.lw_cls * p.ds_w / p.B * w / st[0];
    const float* x = p.cls + idx * p.C1;
    float* dx = p.dcls + idx * p.C1;
    const float lz = p.logz[idx];
    for (int c = 0; c < p.C1; ++c) dx[c] = kc * (expf(x[c] - lz) - (c == target ? 1.f : 0.f));
    // ---- boxes: lw_box * ds_w / (n_has * cnt) * sum over matched GTs of d diou / d box ----
    float gc[3] = {0.f, 0.f, 0.f}, gs[3] = {0.f, 0.f, 0.f};
    if (m) {
        const float kb = p.lw_box * p.ds_w / (fmaxf(p.scale[l * 2 + 1], 1.f) * st[2]);
        const Box6 pb = corners(p.box + idx * 6);
        for (unsigned long long mmask = m; mmask; mmask &= mmask - 1) {
            const int j = __ffsll((long long)mmask) - 1;
            const Box6 tb = corners(p.gt_boxes + (int64_t)(g0 + j) * 6);
            float wh[3], dd[3], lo[3], hi[3];
            float inter = 1.f, vp = 1.f, vt = 1.f, r2 = 0.f, c2 = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo[a] = fmaxf(pb.p1[a], tb.p1[a]); hi[a] = fminf(pb.p2[a], tb.p2[a]);
                wh[a] = fmaxf(hi[a] - lo[a], 0.f);
                dd[a] = pb.p2[a] - pb.p1[a];
                inter *= wh[a]; vp *= dd[a]; vt *= tb.p2[a] - tb.p1[a];
                const float d = (pb.p1[a] + pb.p2[a]) / 2 - (tb.p1[a] + tb.p2[a]) / 2;
                const float e = fminf(pb.p1[a], tb.p1[a]) - fmaxf(pb.p2[a], tb.p2[a]);
                r2 += d * d; c2 += e * e;
            }
            const float un_raw = vp + vt - inter;
            const float un = fmaxf(un_raw, 1e-6f);
            const float un_live = pick_gt(un_raw, 1e-6f);          // union = max(., eps): gradient reaches the volumes only when not clamped
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int a1 = (a + 1) % 3, a2 = (a + 2) % 3;
                const float clamp_live = (hi[a] - lo[a]) >= 0.f ? 1.f : 0.f;            // clamp(min=0) passes the gradient for x >= 0
                const float di_lo = -wh[a1] * wh[a2] * clamp_live, di_hi = wh[a1] * wh[a2] * clamp_live;     // d inter / d lo_a, d hi_a
                const float di1 = di_lo * pick_gt(pb.p1[a], tb.p1[a]), di2 = di_hi * pick_lt(pb.p2[a], tb.p2[a]);
                const float dv1 = -dd[a1] * dd[a2], dv2 = dd[a1] * dd[a2];
                const float du1 = un_live * (dv1 - di1), du2 = un_live * (dv2 - di2);
                const float diou1 = (di1 * un - inter * du1) / (un * un), diou2 = (di2 * un - inter * du2) / (un * un);
                const float d = (pb.p1[a] + pb.p2[a]) / 2 - (tb.p1[a] + tb.p2[a]) / 2;
                const float e = fminf(pb.p1[a], tb.p1[a]) - fmaxf(pb.p2[a], tb.p2[a]);
                const float dr = d;                                                            // d r2 / d p1_a = d r2 / d p2_a
                const float dc1 = 2.f * e * pick_lt(pb.p1[a], tb.p1[a]), dc2 = -2.f * e * pick_gt(pb.p2[a], tb.p2[a]);
                const float g1 = -diou1 + (dr * c2 - r2 * dc1) / (c2 * c2);
                const float g2 = -diou2 + (dr * c2 - r2 * dc2) / (c2 * c2);
                gc[a] += g1 + g2;                      // p1 = c - s/2, p2 = c + s/2
                gs[a] += (g2 - g1) / 2;
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) { gc[a] *= kb; gs[a] *= kb; }
    }
    float* db = p.dbox + idx * 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) { db[a] = gc[a]; db[3 + a] = gs[a]; }
}

// ---- box decode of a yaw-free head: PredBBox's exp + _bbox_pred_to_bbox (unidet3d/encoder.py:99-111, :241-271) in one pass ----
// raw [M][8] (Linear output; the two angle columns are unused here), centres [M][3] -> box [M][6]:
//   e_i = exp(raw_i), i < 6;  centre_a = c_a + (e_{2a+1} - e_{2a}) / 2;  size_a = e_{2a} + e_{2a+1}
__global__ __launch_bounds__(256) void box_decode_fwd_k(const float* __restrict__ raw, const float* __restrict__ cen, int64_t M, float* __restrict__ box) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const float4 r0 = reinterpret_cast<const float4*>(raw + i * 8)[0], r1 = reinterpret_cast<const float4*>(raw + i * 8)[1];
    const float e[6] = {expf(r0.x), expf(r0.y), expf(r0.z), expf(r0.w), expf(r1.x), expf(r1.y)};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        box[i * 6 + a] = cen[i * 3 + a] + (e[2 * a + 1] - e[2 * a]) / 2;
        box[i * 6 + 3 + a] = e[2 * a] + e[2 * a + 1];
    }
}
__global__ __launch_bounds__(256) void box_decode_bwd_k(const float* __restrict__ raw, const float* __restrict__ dbox, int64_t M, float* __restrict__ draw) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const float4 r0 = reinterpret_cast<const float4*>(raw + i * 8)[0], r1 = reinterpret_cast<const float4*>(raw + i * 8)[1];
    const float e[6] = {expf(r0.x), expf(r0.y), expf(r0.z), expf(r0.w), expf(r1.x), expf(r1.y)};
    float g[8];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float dc = dbox[i * 6 + a], ds = dbox[i * 6 + 3 + a];
        g[2 * a] = e[2 * a] * (ds - dc / 2);
        g[2 * a + 1] = e[2 * a + 1] * (ds + dc / 2);
    }
    g[6] = 0.f; g[7] = 0.f;
    reinterpret_cast<float4*>(draw + i * 8)[0] = make_float4(g[0], g[1], g[2], g[3]);
    reinterpret_cast<float4*>(draw + i * 8)[1] = make_float4(g[4], g[5], g[6], g[7]);
}

}  // namespace u3d

using namespace u3d;

extern "C" {

int u3d_box_decode_fwd(const float* raw, const float* centers, int64_t M, float* box, u3d_stream_t stream) {
    if (!raw || !centers || !box || M < 0) return U3D_EINVAL;
    if (M == 0) return U3D_OK;
    hipLaunchKernelGGL(box_decode_fwd_k, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, (hipStream_t)stream, raw, centers, M, box);
    return check_launch("box_decode_fwd");
}

int u3d_box_decode_bwd(const float* raw, const float* dbox, int64_t M, float* draw, u3d_stream_t stream) {
    if (!raw || !dbox || !draw || M < 0) return U3D_EINVAL;
    if (M == 0) return U3D_OK;
    hipLaunchKernelGGL(box_decode_bwd_k, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, (hipStream_t)stream, raw, dbox, M, draw);
    return check_launch("box_decode_bwd");
}

static inline int64_t al64(int64_t x) { return (x + 63) & ~(int64_t)63; }

int64_t u3d_criterion_ws_bytes(int L, int B, int64_t n_tot, int64_t G, int64_t P) {
    return al64((int64_t)L * P * 4) + al64((int64_t)L * n_tot * 4) + al64((int64_t)L * G * 4) + al64((int64_t)L * n_tot * 8) +
           al64((int64_t)L * B * 16) + al64((int64_t)L * 8) + 256;
}

int u3d_criterion_packed(const float* cls, const float* box, const int32_t* cu, const int32_t* gt_off, const int64_t* gt_labels,
                         const float* gt_boxes, const uint8_t* qmask, const int64_t* qm_off, int L, int B, int64_t n_tot, int C1, int64_t G,
                         int64_t P, int max_gt, int min_queries_with_gt, int topk, float w_cls, float w_box, float non_obj_w, float ds_w,
                         float lw_cls, float lw_box, float* loss, float* dcls, float* dbox, void* ws, u3d_stream_t stream) {
    if (!cls || !box || !cu || !gt_off || !qm_off || !loss || !dcls || !dbox || !ws || L <= 0 || B <= 0 || n_tot <= 0 || C1 < 2 || G < 0 || P < 0)
        return U3D_EINVAL;
    if (G > 0 && (!gt_labels || !gt_boxes || !qmask)) return U3D_EINVAL;
    if (max_gt > 64) { set_error("criterion: %d ground-truth boxes in one scene exceed the 64-bit match mask", max_gt); return U3D_EUNSUPPORTED; }
    if (G > 0 && min_queries_with_gt < topk + 1) {        // torch.topk(cost, topk + 1, dim=0) of the reference raises here as well
        set_error("criterion: a scene with ground truth has %d queries, fewer than topk + 1 = %d", min_queries_with_gt, topk + 1);
        return U3D_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    CritParams p;
    p.cls = cls; p.box = box; p.cu = cu; p.gt_off = gt_off; p.gt_labels = gt_labels; p.gt_boxes = gt_boxes; p.qmask = qmask; p.qm_off = qm_off;
    p.L = L; p.B = B; p.C1 = C1; p.topk = topk; p.n_tot = n_tot; p.G = G; p.P = P;
    p.w_cls = w_cls; p.w_box = w_box; p.non_obj_w = non_obj_w; p.ds_w = ds_w; p.lw_cls = lw_cls; p.lw_box = lw_box;
    char* w = (char*)ws;
    p.cost = (float*)w; w += al64((int64_t)L * P * 4);
    p.logz = (float*)w; w += al64((int64_t)L * n_tot * 4);
    p.kth = (float*)w; w += al64((int64_t)L * G * 4);
    p.mm = (unsigned long long*)w; w += al64((int64_t)L * n_tot * 8);
    p.stats = (float*)w; w += al64((int64_t)L * B * 16);
    p.scale = (float*)w;
    p.loss = loss; p.dcls = dcls; p.dbox = dbox;
    const unsigned gq = (unsigned)ceil_div((int64_t)L * n_tot, 256);
    hipLaunchKernelGGL(crit_cost_k, dim3(gq), dim3(256), 0, s, p);
    if (G > 0) hipLaunchKernelGGL(crit_kth_k, dim3((unsigned)((int64_t)L * G)), dim3(64), 0, s, p);
    hipLaunchKernelGGL(crit_stats_k, dim3((unsigned)(L * B)), dim3(256), 0, s, p);
    hipLaunchKernelGGL(crit_final_k, dim3(1), dim3(1), 0, s, p);
    hipLaunchKernelGGL(crit_grad_k, dim3(gq), dim3(256), 0, s, p);
    return check_launch("criterion_packed");
}

}  // extern "C"
