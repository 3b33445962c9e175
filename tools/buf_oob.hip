// Does the raw-buffer bounds check include the scalar offset?  Measured on MI355X (gfx950), round 6: YES -- the load with the
// scalar offset past num_records returns 0.0 like the one with the vector offset past it ("soffset past num_records -> 0.0").
// (An earlier version of this header claimed the opposite from the GFX9 ISA text; the kernels that bound rows through the scalar
// offset -- gemm.hip nt_epilogue, the spconv write-outs -- rely on what the probe prints.)
// build: hipcc --offload-arch=gfx950 -O3 tools/buf_oob.hip -o tools/bin/buf_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* src, float* out, int soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 64, 0x00020000);      // 16 floats visible
    out[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 0, soff, 0));          // voffset 0, soffset past the end
    out[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, soff, 0, 0));          // voffset past the end
    out[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, 60, 0, 0));            // last valid dword
}
int main() {
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = 100.f + i;
    float *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 16); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, o, 128);
    float r[3]; hipMemcpy(r, o, 12, hipMemcpyDeviceToHost);
    printf("soffset past num_records -> %.1f (132 = NOT bounds-checked, 0 = checked); voffset past -> %.1f; last valid -> %.1f\n", r[0], r[1], r[2]);
    return 0;
}
