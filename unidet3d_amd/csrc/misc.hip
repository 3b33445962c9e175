// Error reporting, per-class event timing, device-wide exclusive scan (wave-64 shuffle scan).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "u3d_common.h"

namespace u3d {

static thread_local char g_err[512] = "";

int g_fp32_math = -1;
int g_conv_kernel = -1;
bool fp32_x3() {
    if (g_fp32_math < 0) {
        const char* e = getenv("U3D_FP32_MATH");
        g_fp32_math = (e && !strcmp(e, "mfma")) ? 0 : 1;
    }
    return g_fp32_math == 1;
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------------------------------
struct ProfClass {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    double work = 0.0;
};
static ProfClass g_prof[U3D_K_COUNT];
static std::mutex g_prof_mu;

ProfScope::ProfScope(int c, hipStream_t s, double w) : cls(c), stream(s), on(false), work(w) {
    if (c < 0 || c >= U3D_K_COUNT || !g_prof[c].on) return;
    on = true;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, stream);
}
ProfScope::~ProfScope() {
    if (!on) return;
    hipEventRecord(e1, stream);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof[cls].ev.emplace_back(e0, e1);
    g_prof[cls].work += work;
}

// ------------------------------------------------------------------------------------------
// scan: 256 threads x 8 items per block
constexpr int SCAN_T = 256, SCAN_I = 8, SCAN_B = SCAN_T * SCAN_I;

struct LoadPopc {
    const uint64_t* p;
    __device__ int operator()(int64_t i) const { return __popcll(p[i]); }
};
struct LoadI32 {
    const int32_t* p;
    __device__ int operator()(int64_t i) const { return p[i]; }
};

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// block-wide exclusive scan of per-thread sums; returns exclusive prefix of this thread, total via *tot
__device__ __forceinline__ int block_excl_scan(int tsum, int* s_wave /*[4+1]*/, int* tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = wave_incl_scan(tsum, lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_T / 64; ++w) {
        int v = s_wave[w];
        if (w < wave) woff += v;
        total += v;
    }
    __syncthreads();
    *tot = total;
    return woff + incl - tsum;
}

template <class F>
__global__ __launch_bounds__(SCAN_T) void scan_reduce_k(F f, int64_t n, int32_t* block_sums) {
    __shared__ int s_wave[8];
    const int64_t base = (int64_t)blockIdx.x * SCAN_B + (int64_t)threadIdx.x * SCAN_I;
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i)
        if (base + i < n) s += f(base + i);
    int tot;
    block_excl_scan(s, s_wave, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// out[i] = block_off[blk] + exclusive prefix inside the block, for i in [0, n]; f(n) == 0
template <class F>
__global__ __launch_bounds__(SCAN_T) void scan_down_k(F f, int64_t n, const int32_t* block_off, int32_t* out) {
    __shared__ int s_wave[8];
    const int64_t base = (int64_t)blockIdx.x * SCAN_B + (int64_t)threadIdx.x * SCAN_I;
    int v[SCAN_I];
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) {
        v[i] = (base + i < n) ? f(base + i) : 0;
        s += v[i];
    }
    int tot;
    int ex = block_excl_scan(s, s_wave, &tot) + (block_off ? block_off[blockIdx.x] : 0);
#pragma unroll
    for (int i = 0; i < SCAN_I; ++i) {
        if (base + i <= n) out[base + i] = ex;
        ex += v[i];
    }
}

int64_t scan_ws_bytes(int64_t n) {
    int64_t bytes = 0;
    int64_t m = n + 1;
    while (m > SCAN_B) {
        int64_t nb = ceil_div(m, SCAN_B);
        bytes += (nb + (nb + 1) + 8) * 4;
        m = nb + 1;
    }
    return bytes + 256;
}

template <class F>
static int scan_impl(F f, int64_t n, int32_t* out, void* ws, hipStream_t s) {
    const int64_t m = n + 1;   // we also write out[n]
    const int64_t nb = ceil_div(m, SCAN_B);
    if (nb == 1) {
        hipLaunchKernelGGL(scan_down_k<F>, dim3(1), dim3(SCAN_T), 0, s, f, n, (const int32_t*)nullptr, out);
        return check_launch("scan_down");
    }
    int32_t* sums = (int32_t*)ws;
    int32_t* offs = sums + nb;
    void* ws2 = (void*)(offs + nb + 1 + 8);
    hipLaunchKernelGGL(scan_reduce_k<F>, dim3((unsigned)nb), dim3(SCAN_T), 0, s, f, n, sums);
    int rc = check_launch("scan_reduce");
    if (rc) return rc;
    rc = scan_impl(LoadI32{sums}, nb, offs, ws2, s);
    if (rc) return rc;
    hipLaunchKernelGGL(scan_down_k<F>, dim3((unsigned)nb), dim3(SCAN_T), 0, s, f, n, (const int32_t*)offs, out);
    return check_launch("scan_down");
}

int exclusive_scan_popc64(const uint64_t* words, int64_t n, int32_t* out, void* ws, hipStream_t s) {
    return scan_impl(LoadPopc{words}, n, out, ws, s);
}
int exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, void* ws, hipStream_t s) {
    return scan_impl(LoadI32{in}, n, out, ws, s);
}

}  // namespace u3d

extern "C" {

int u3d_version(void) { return U3D_ABI_VERSION; }
int u3d_conv_kernel(int mode) {
    if (u3d::g_conv_kernel < 0) {
        const char* e = getenv("U3D_GMM_WG");
        u3d::g_conv_kernel = (e && atoi(e) == 0) ? 0 : 1;
    }
    const int prev = u3d::g_conv_kernel;
    if (mode == 0 || mode == 1 || mode == 2) u3d::g_conv_kernel = mode;
    return prev;
}
int u3d_fp32_math(int mode) {
    const int prev = u3d::fp32_x3() ? 1 : 0;
    if (mode == 0 || mode == 1) u3d::g_fp32_math = mode;
    return prev;
}
const char* u3d_last_error(void) { return u3d::g_err; }

int u3d_prof_enable(int c, int on) {
    if (c < 0 || c >= U3D_K_COUNT) return U3D_EINVAL;
    std::lock_guard<std::mutex> lk(u3d::g_prof_mu);
    u3d::g_prof[c].on = on != 0;
    return U3D_OK;
}

int u3d_prof_collect(int c, double* total_ms, int64_t* launches, double* work) {
    if (c < 0 || c >= U3D_K_COUNT) return U3D_EINVAL;
    std::lock_guard<std::mutex> lk(u3d::g_prof_mu);
    auto& pc = u3d::g_prof[c];
    double ms = 0.0;
    for (auto& e : pc.ev) {
        hipEventSynchronize(e.second);
        float t = 0.f;
        hipEventElapsedTime(&t, e.first, e.second);
        ms += t;
        hipEventDestroy(e.first);
        hipEventDestroy(e.second);
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = (int64_t)pc.ev.size();
    if (work) *work = pc.work;
    pc.ev.clear();
    pc.work = 0.0;
    return U3D_OK;
}

int64_t u3d_index_rank_ws_bytes(int64_t n_words) { return u3d::scan_ws_bytes(n_words); }

int u3d_index_rank(const uint64_t* bitmap, int64_t n_words, int32_t* word_rank, void* ws, u3d_stream_t stream) {
    if (!bitmap || !word_rank || n_words <= 0) return U3D_EINVAL;
    return u3d::exclusive_scan_popc64(bitmap, n_words, word_rank, ws, (hipStream_t)stream);
}

}  // extern "C"
