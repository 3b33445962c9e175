// K-post: inference post-processing of one scene on the device:
//   * class-wise greedy NMS of axis-aligned boxes -- unidet3d/unidet3d.py:595-650: fast_nms=True is
//     mmcv.ops.nms3d_normal (mmcv @780ffed, ops/csrc/common/cuda/iou3d_cuda_kernel.cuh `iou_normal`: IoU of the
//     (x, y, dx, dy) rectangles, z and heading ignored), fast_nms=False is mmdet3d 1.4.0 aligned_3d_nms (3-D IoU of
//     the corner boxes); both visit a class in descending score order;
//   * superpoint trimming of the surviving boxes -- unidet3d/unidet3d.py:540-593 + get_face_distances :652-677:
//     point-in-box test, per-superpoint inside ratio, delete (< low) / add (> up) whole superpoints, min/max of
//     the selected points.
// Both are integer / comparison work on small inputs: bit-exact against oracle/postproc.py, which repeats the same
// fp32 operation order (this file is built with -ffp-contract=off so no multiply-add is fused).
#include <math.h>

#include "u3d_common.h"

namespace u3d {

constexpr int NMS_MAX = 4400;          // 9 LDS words per box: 158 KB of the CU's 160 KB (launches above 64 KB opt in, see lds_opt_in)
constexpr int NMS_ROT_MAX = 3600;      // 11 LDS words per box

// MODE 0: BEV IoU of (cx, cy, cz, dx, dy, dz) boxes, suppress when iou > thr           (mmcv nms3d_normal / iou_normal)
// MODE 1: 3-D IoU of (x1, y1, z1, x2, y2, z2) boxes, survive only when iou <= thr       (mmdet3d aligned_3d_nms: a 0/0 IoU
//         of two zero-volume boxes is NaN there and NaN <= thr is false, so such a box is dropped)
template <int MODE>
__global__ __launch_bounds__(1024) void nms_k(const float* __restrict__ boxes, const int32_t* __restrict__ labels, int n, float thr,
                                              uint8_t* __restrict__ keep) {
    extern __shared__ float sm[];
    float* x1 = sm; float* x2 = x1 + n; float* y1 = x2 + n; float* y2 = y1 + n; float* z1 = y2 + n; float* z2 = z1 + n; float* ar = z2 + n;
    int* lab = reinterpret_cast<int*>(ar + n);
    int* sup = lab + n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float* b = boxes + i * 6;
        if (MODE == 0) {
            x1[i] = b[0] - b[3] / 2; x2[i] = b[0] + b[3] / 2; y1[i] = b[1] - b[4] / 2; y2[i] = b[1] + b[4] / 2; z1[i] = 0.f; z2[i] = 0.f;
            ar[i] = b[3] * b[4];
        } else {
            x1[i] = b[0]; y1[i] = b[1]; z1[i] = b[2]; x2[i] = b[3]; y2[i] = b[4]; z2[i] = b[5];
            ar[i] = (b[3] - b[0]) * (b[4] - b[1]) * (b[5] - b[2]);
        }
        lab[i] = labels[i]; sup[i] = 0;
        keep[i] = 0;
    }
    for (int i = 0; i < n; ++i) {
        __syncthreads();
        if (sup[i]) continue;                     // same LDS word for every thread: uniform
        if (threadIdx.x == 0) keep[i] = 1;
        const int li = lab[i];
        const float ax1 = x1[i], ax2 = x2[i], ay1 = y1[i], ay2 = y2[i], az1 = z1[i], az2 = z2[i], sa = ar[i];
        for (int j = i + 1 + threadIdx.x; j < n; j += blockDim.x) {
            if (lab[j] != li) break;              // labels ascending: the class segment ended
            if (sup[j]) continue;
            const float w = fmaxf(fminf(ax2, x2[j]) - fmaxf(ax1, x1[j]), 0.f), h = fmaxf(fminf(ay2, y2[j]) - fmaxf(ay1, y1[j]), 0.f);
            if (MODE == 0) {
                const float inter = w * h;
                const float iou = inter / fmaxf(sa + ar[j] - inter, 1e-8f);
                if (iou > thr) sup[j] = 1;
            } else {
                const float d = fmaxf(fminf(az2, z2[j]) - fmaxf(az1, z1[j]), 0.f);
                const float inter = w * h * d;
                const float iou = inter / (sa + ar[j] - inter);
                if (!(iou <= thr)) sup[j] = 1;
            }
        }
    }
}

// ---- rotated boxes: mmcv.ops.nms3d (unidet3d.py:626, with_yaw) -- greedy suppression by the BEV IoU of the rotated
// rectangles (x, y, dx, dy, heading).  The intersection area is summed edge by edge (no vertex sorting, no arrays): every
// edge of A is clipped to the inside of B (closed), every edge of B to the strict inside of A (so an edge shared by both
// outlines counts once), and a clipped piece P0->P1 of a counter-clockwise outline contributes cross(P0, P1) / 2.
__device__ __forceinline__ float clipped_edges_area(const float (&pa)[8], const float (&pb)[8], bool strict) {
    float area = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float px = pa[2 * e], py = pa[2 * e + 1];
        const float dx = pa[2 * ((e + 1) & 3)] - px, dy = pa[2 * ((e + 1) & 3) + 1] - py;
        float t0 = 0.f, t1 = 1.f;
        bool alive = true;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const float ax = pb[2 * f], ay = pb[2 * f + 1];
            const float ex = pb[2 * ((f + 1) & 3)] - ax, ey = pb[2 * ((f + 1) & 3) + 1] - ay;
            const float n0 = ex * (py - ay) - ey * (px - ax);       // side of the edge's start, > 0 = inside
            const float m = ex * dy - ey * dx;                       // change of the side along the edge
            if (m == 0.f) {
                alive = alive && (strict ? n0 > 0.f : n0 >= 0.f);
            } else {
                const float tc = -n0 / m;
                if (m > 0.f) t0 = fmaxf(t0, tc); else t1 = fminf(t1, tc);
            }
        }
        if (alive && t0 < t1) {
            const float x0 = px + t0 * dx, y0 = py + t0 * dy, x1 = px + t1 * dx, y1 = py + t1 * dy;
            area += 0.5f * (x0 * y1 - x1 * y0);
        }
    }
    return area;
}

__global__ __launch_bounds__(1024) void nms_rot_k(const float* __restrict__ boxes, const int32_t* __restrict__ labels, int n, float thr,
                                                  uint8_t* __restrict__ keep) {
    extern __shared__ float sm[];
    float* cor = sm;                     // [n][8] corners, counter-clockwise
    float* ar = cor + 8 * n;
    int* lab = reinterpret_cast<int*>(ar + n);
    int* sup = lab + n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float* b = boxes + i * 7;
        const float c = cosf(b[6]), s = sinf(b[6]), hx = 0.5f * b[3], hy = 0.5f * b[4];
        const float sx[4] = {hx, -hx, -hx, hx}, sy[4] = {hy, hy, -hy, -hy};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            cor[i * 8 + 2 * v] = sx[v] * c - sy[v] * s + b[0];
            cor[i * 8 + 2 * v + 1] = sx[v] * s + sy[v] * c + b[1];
        }
        ar[i] = b[3] * b[4];
        lab[i] = labels[i]; sup[i] = 0;
        keep[i] = 0;
    }
    for (int i = 0; i < n; ++i) {
        __syncthreads();
        if (sup[i]) continue;
        if (threadIdx.x == 0) keep[i] = 1;
        const int li = lab[i];
        const float ox = cor[i * 8], oy = cor[i * 8 + 1];           // coordinates relative to a corner of box i
        float pa[8];
#pragma unroll
        for (int v = 0; v < 4; ++v) { pa[2 * v] = cor[i * 8 + 2 * v] - ox; pa[2 * v + 1] = cor[i * 8 + 2 * v + 1] - oy; }
        const float sa = ar[i];
        for (int j = i + 1 + threadIdx.x; j < n; j += blockDim.x) {
            if (lab[j] != li) break;
            if (sup[j]) continue;
            float pb[8];
#pragma unroll
            for (int v = 0; v < 4; ++v) { pb[2 * v] = cor[j * 8 + 2 * v] - ox; pb[2 * v + 1] = cor[j * 8 + 2 * v + 1] - oy; }
            const float inter = fmaxf(clipped_edges_area(pa, pb, false) + clipped_edges_area(pb, pa, true), 0.f);
            const float iou = inter / fmaxf(sa + ar[j] - inter, 1e-8f);
            if (iou > thr) sup[j] = 1;
        }
    }
}

__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
    if (v >= 0.f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ void trim_init_k(float* __restrict__ mm, int nb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nb * 6) mm[i] = (i % 6) < 3 ? INFINITY : -INFINITY;
}

// one wave per (superpoint s, tile of 64 boxes): lane = box.  The superpoint's points are contiguous in the CSR list,
// so the wave knows the inside ratio of (box, s) after one pass and can apply the delete / add rule right away.
__global__ __launch_bounds__(64) void trim_k(const float* __restrict__ points, int64_t ld, const int32_t* __restrict__ list,
                                             const int32_t* __restrict__ offsets, int nbt, const float* __restrict__ boxes, int nb,
                                             int box_dim, float low, float up, float* __restrict__ mm) {
    const int s = blockIdx.x / nbt, b = (blockIdx.x % nbt) * 64 + threadIdx.x;
    const int o0 = offsets[s], o1 = offsets[s + 1];
    if (o0 >= o1 || b >= nb) return;
    const float cx = boxes[b * box_dim + 0], cy = boxes[b * box_dim + 1], cz = boxes[b * box_dim + 2];
    const float hx = boxes[b * box_dim + 3] / 2, hy = boxes[b * box_dim + 4] / 2, hz = boxes[b * box_dim + 5] / 2;
    // heading (7-dof boxes): the shift p - c is rotated by -yaw about z before the face test (get_face_distances :666-668,
    // mmdet3d rotation_3d_in_axis: x' = x cos a - y sin a, y' = x sin a + y cos a with a = -yaw)
    const bool rot = box_dim == 7;
    float rs = 0.f, rc = 1.f;
    if (rot) { rs = sinf(-boxes[b * 7 + 6]); rc = cosf(-boxes[b * 7 + 6]); }
    float imin[3] = {INFINITY, INFINITY, INFINITY}, imax[3] = {-INFINITY, -INFINITY, -INFINITY};     // inside points
    float amin[3] = {INFINITY, INFINITY, INFINITY}, amax[3] = {-INFINITY, -INFINITY, -INFINITY};     // all points of s
    int cnt_in = 0;
    for (int o = o0; o < o1; ++o) {
        const float* p = points + (int64_t)list[o] * ld;
        const float px = p[0], py = p[1], pz = p[2];
        // get_face_distances with yaw 0: shift = p - c; centre' = c + shift; distances to the six faces
        float sx = px - cx, sy = py - cy;
        if (rot) { const float tx = sx * rc - sy * rs; sy = sx * rs + sy * rc; sx = tx; }
        const float ex = cx + sx, ey = cy + sy, ez = cz + (pz - cz);
        const bool in = ((ex - cx) + hx > 0.f) && ((cx + hx) - ex > 0.f) && ((ey - cy) + hy > 0.f) && ((cy + hy) - ey > 0.f) &&
                        ((ez - cz) + hz > 0.f) && ((cz + hz) - ez > 0.f);
        amin[0] = fminf(amin[0], px); amin[1] = fminf(amin[1], py); amin[2] = fminf(amin[2], pz);
        amax[0] = fmaxf(amax[0], px); amax[1] = fmaxf(amax[1], py); amax[2] = fmaxf(amax[2], pz);
        if (in) {
            ++cnt_in;
            imin[0] = fminf(imin[0], px); imin[1] = fminf(imin[1], py); imin[2] = fminf(imin[2], pz);
            imax[0] = fmaxf(imax[0], px); imax[1] = fmaxf(imax[1], py); imax[2] = fmaxf(imax[2], pz);
        }
    }
    const float ratio = (float)cnt_in / (float)(o1 - o0);      // scatter_mean of the 0/1 inside flags
    const bool add = ratio > up, del = ratio < low;            // :574-578: delete first, then add
    if (!add && (del || cnt_in == 0)) return;
    float* out = mm + b * 6;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        atomic_min_f32(out + d, add ? amin[d] : imin[d]);
        atomic_max_f32(out + 3 + d, add ? amax[d] : imax[d]);
    }
}

}  // namespace u3d

using namespace u3d;

extern "C" {

// a launch that wants more than 64 KB of dynamic LDS has to raise the kernel's limit first (gfx950: 160 KB per workgroup)
static int lds_opt_in(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return U3D_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        (void)hipGetLastError();
        set_error("nms: cannot reserve %zu bytes of LDS", bytes);
        return U3D_EUNSUPPORTED;
    }
    return U3D_OK;
}

static int launch_nms(int mode, const float* boxes, const int32_t* labels, int n, float iou_thr, uint8_t* keep, u3d_stream_t stream) {
    if (n < 0 || (n > 0 && (!boxes || !labels || !keep))) return U3D_EINVAL;
    if (n == 0) return U3D_OK;
    if (n > NMS_MAX) {
        set_error("nms: %d boxes exceed the single-workgroup limit of %d", n, NMS_MAX);
        return U3D_EUNSUPPORTED;
    }
    const size_t lds = (size_t)n * 9 * sizeof(float);
    if (int rc = lds_opt_in(mode == 0 ? (const void*)nms_k<0> : (const void*)nms_k<1>, lds)) return rc;
    if (mode == 0) hipLaunchKernelGGL(nms_k<0>, dim3(1), dim3(1024), lds, (hipStream_t)stream, boxes, labels, n, iou_thr, keep);
    else hipLaunchKernelGGL(nms_k<1>, dim3(1), dim3(1024), lds, (hipStream_t)stream, boxes, labels, n, iou_thr, keep);
    return check_launch("nms");
}

int u3d_nms_bev(const float* boxes, const int32_t* labels, int n, float iou_thr, uint8_t* keep, u3d_stream_t stream) {
    return launch_nms(0, boxes, labels, n, iou_thr, keep, stream);
}

int u3d_nms_rotated(const float* boxes, const int32_t* labels, int n, float iou_thr, uint8_t* keep, u3d_stream_t stream) {
    if (n < 0 || (n > 0 && (!boxes || !labels || !keep))) return U3D_EINVAL;
    if (n == 0) return U3D_OK;
    if (n > NMS_ROT_MAX) {
        set_error("nms_rotated: %d boxes exceed the single-workgroup limit of %d", n, NMS_ROT_MAX);
        return U3D_EUNSUPPORTED;
    }
    if (int rc = lds_opt_in((const void*)nms_rot_k, (size_t)n * 11 * sizeof(float))) return rc;
    hipLaunchKernelGGL(nms_rot_k, dim3(1), dim3(1024), (size_t)n * 11 * sizeof(float), (hipStream_t)stream, boxes, labels, n, iou_thr, keep);
    return check_launch("nms_rotated");
}

int u3d_nms_aligned3d(const float* corners, const int32_t* labels, int n, float iou_thr, uint8_t* keep, u3d_stream_t stream) {
    return launch_nms(1, corners, labels, n, iou_thr, keep, stream);
}

int u3d_trim_boxes(const float* points, int64_t pt_ld, const int32_t* sp_list, const int32_t* sp_offsets, int S,
                   const float* boxes, int nb, int box_dim, float low_thr, float up_thr, float* minmax, u3d_stream_t stream) {
    if (nb < 0 || S < 0 || pt_ld < 3 || (box_dim != 6 && box_dim != 7) || (nb > 0 && (!boxes || !minmax)) || (S > 0 && (!points || !sp_list || !sp_offsets))) return U3D_EINVAL;
    if (nb == 0) return U3D_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(trim_init_k, dim3((unsigned)ceil_div(nb * 6, 256)), dim3(256), 0, s, minmax, nb);
    if (S > 0) {
        const int nbt = (int)ceil_div(nb, 64);
        hipLaunchKernelGGL(trim_k, dim3((unsigned)((int64_t)S * nbt)), dim3(64), 0, s, points, pt_ld, sp_list, sp_offsets, nbt, boxes, nb, box_dim,
                           low_thr, up_thr, minmax);
    }
    return check_launch("trim_boxes");
}

}  // extern "C"
