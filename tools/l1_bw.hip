// Micro-benchmark: per-CU load throughput of global_load_dwordx4 on gfx950 for L1-resident, L2-resident and
// gather-shaped access (the sparse-conv fragment pattern: 16 rows x 64 B per wave instruction).
// build: hipcc --offload-arch=gfx950 -O3 tools/l1_bw.hip -o tools/bin/l1_bw
#include <hip/hip_runtime.h>
#include <stdio.h>

// mode 0: coalesced 1 KB per wave load, footprint `span` bytes per wave-stream (wraps)
// mode 1: gather: lane (i16,q) reads 16 B at row (r0 + i16*stride_rows) * 128 + q*16  (+64 for the 2nd half)
__global__ __launch_bounds__(256) void k(const float4* __restrict__ buf, float* out, int iters, long span16, int mode, long wave_off16) {
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float4 acc = make_float4(0, 0, 0, 0);
    long base = (wid * wave_off16) % span16;
    for (int i = 0; i < iters; ++i) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            long idx;
            if (mode == 0) idx = (base + u * 64 + lane) % span16;
            else if (mode == 1) idx = (base + (long)(lane & 15) * 8 * 37 + (u & 1) * 4 + (lane >> 4) + (u >> 1) * 8 * 1021) % span16;   // 128-B rows, scattered
            else idx = (base + (long)((lane >> 3) + (u & 1) * 8) * 8 * 37 + (lane & 7) + (u >> 1) * 8 * 1021) % span16;   // mode 2: 8 COMPLETE 128-B rows per instruction
            v[u] = buf[idx];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        base = (base + 512 + 8 * 4099) % span16;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    const long bytes = 1l << 30;
    float4* buf; float* out;
    hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes); hipMalloc(&out, 8192 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400;
    struct { const char* name; long span; int mode; long woff; } cases[] = {
        {"coalesced, 16 KB footprint (L1 hits)", 16l << 10, 0, 0},
        {"coalesced, 16 MB footprint (L2 hits)", 16l << 20, 0, 4096},
        {"coalesced, 1 GB footprint (HBM)", 1l << 30, 0, 65536},
        {"gather 16 rows x 64 B, 16 KB footprint", 16l << 10, 1, 0},
        {"gather 16 rows x 64 B, 16 MB footprint", 16l << 20, 1, 4099},
        {"gather 16 rows x 64 B, 200 MB footprint", 200l << 20, 1, 65537},
        {"gather 8 rows x 128 B, 16 KB footprint", 16l << 10, 2, 0},
        {"gather 8 rows x 128 B, 16 MB footprint", 16l << 20, 2, 4099},
        {"gather 8 rows x 128 B, 200 MB footprint", 200l << 20, 2, 65537},
    };
    for (auto& c : cases)
        for (int wgs : {1024, 4096}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, buf, out, iters, c.span / 16, c.mode, c.woff);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double b = (double)wgs * 4 * iters * 8 * 1024;
            printf("%-44s wgs=%5d %8.3f ms %8.2f TB/s  %6.1f B/clk/CU (2.4 GHz)\n", c.name, wgs, ms, b / ms / 1e9, b / (ms * 1e-3) / 256 / 2.4e9);
        }
    return 0;
}
