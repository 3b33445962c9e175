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

constexpr int NMS_MAX = 1800;          // 9 LDS words per box: 64.8 KB, just inside the 64 KB a launch may request without opting in

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
                                             float low, float up, float* __restrict__ mm) {
    const int s = blockIdx.x / nbt, b = (blockIdx.x % nbt) * 64 + threadIdx.x;
    const int o0 = offsets[s], o1 = offsets[s + 1];
    if (o0 >= o1 || b >= nb) return;
    const float cx = boxes[b * 6 + 0], cy = boxes[b * 6 + 1], cz = boxes[b * 6 + 2];
    const float hx = boxes[b * 6 + 3] / 2, hy = boxes[b * 6 + 4] / 2, hz = boxes[b * 6 + 5] / 2;
    float imin[3] = {INFINITY, INFINITY, INFINITY}, imax[3] = {-INFINITY, -INFINITY, -INFINITY};     // inside points
    float amin[3] = {INFINITY, INFINITY, INFINITY}, amax[3] = {-INFINITY, -INFINITY, -INFINITY};     // all points of s
    int cnt_in = 0;
    for (int o = o0; o < o1; ++o) {
        const float* p = points + (int64_t)list[o] * ld;
        const float px = p[0], py = p[1], pz = p[2];
        // get_face_distances with yaw 0: shift = p - c; centre' = c + shift; distances to the six faces
        const float ex = cx + (px - cx), ey = cy + (py - cy), ez = cz + (pz - cz);
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

static int launch_nms(int mode, const float* boxes, const int32_t* labels, int n, float iou_thr, uint8_t* keep, u3d_stream_t stream) {
    if (n < 0 || (n > 0 && (!boxes || !labels || !keep))) return U3D_EINVAL;
    if (n == 0) return U3D_OK;
    if (n > NMS_MAX) {
        set_error("nms: %d boxes exceed the single-workgroup limit of %d", n, NMS_MAX);
        return U3D_EUNSUPPORTED;
    }
    const size_t lds = (size_t)n * 9 * sizeof(float);
    if (mode == 0) hipLaunchKernelGGL(nms_k<0>, dim3(1), dim3(1024), lds, (hipStream_t)stream, boxes, labels, n, iou_thr, keep);
    else hipLaunchKernelGGL(nms_k<1>, dim3(1), dim3(1024), lds, (hipStream_t)stream, boxes, labels, n, iou_thr, keep);
    return check_launch("nms");
}

int u3d_nms_bev(const float* boxes, const int32_t* labels, int n, float iou_thr, uint8_t* keep, u3d_stream_t stream) {
    return launch_nms(0, boxes, labels, n, iou_thr, keep, stream);
}

int u3d_nms_aligned3d(const float* corners, const int32_t* labels, int n, float iou_thr, uint8_t* keep, u3d_stream_t stream) {
    return launch_nms(1, corners, labels, n, iou_thr, keep, stream);
}

int u3d_trim_boxes(const float* points, int64_t pt_ld, const int32_t* sp_list, const int32_t* sp_offsets, int S,
                   const float* boxes, int nb, float low_thr, float up_thr, float* minmax, u3d_stream_t stream) {
    if (nb < 0 || S < 0 || pt_ld < 3 || (nb > 0 && (!boxes || !minmax)) || (S > 0 && (!points || !sp_list || !sp_offsets))) return U3D_EINVAL;
    if (nb == 0) return U3D_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(trim_init_k, dim3((unsigned)ceil_div(nb * 6, 256)), dim3(256), 0, s, minmax, nb);
    if (S > 0) {
        const int nbt = (int)ceil_div(nb, 64);
        hipLaunchKernelGGL(trim_k, dim3((unsigned)((int64_t)S * nbt)), dim3(64), 0, s, points, pt_ld, sp_list, sp_offsets, nbt, boxes, nb, low_thr,
                           up_thr, minmax);
    }
    return check_launch("trim_boxes");
}

}  // extern "C"
