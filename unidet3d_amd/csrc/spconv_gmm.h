// Shared between the two output-stationary sparse-convolution kernels (spconv.hip: wave-private tiles, spconv_wg.hip:
// workgroup tiles with the weights of an offset shared through LDS).  Internal, not part of the C ABI.
#pragma once
#include "u3d_common.h"

namespace u3d {

struct GmmParams {
    const float* src;
    const float* w;
    const int32_t* gather;
    const int32_t* scatter;
    const int32_t* ts;
    const float* addend;
    float* out;          // dst, or the partial buffer [G][n_dst][Cd] when G > 1
    float* stats;        // nullable (G == 1 only): per-tile column sums of dst for the batch norm behind this convolution,
                         // float [n_sub][2][Cd] = sum x | sum x^2 over the tile's rows
    int K;
    int64_t cap;
    int Cs, Cd;
    int64_t n_dst;
    int64_t n_src;
    int64_t n_sub;
    int n_slices;
    int G;
    int kper;
};

constexpr int GMM_CDS = 32;            // output columns per wave
// Accumulator tile of a wave in LDS: R rows x 32 floats.  Default layout (round 3): 128-byte rows, the eight 16-byte quads of
// row r stored at quad ^ (r & 7) -- the XOR spreads the random-row 16-byte accesses of the MFMA read-modify-write over the
// banks like the padded 160-byte rows of rounds 1-2 did, without their 25 % padding, and the scratch row for lanes past the
// end of a range aliases the head of the staging image (dead at that point of a unit) instead of being a 65th row:
// 10 KB per wave instead of 12.4 -> FOUR workgroups (16 waves) per CU instead of three for the 64-row kernels.
// -DU3D_GMM_ALD40 builds the old layout (A/B measurements).
#ifdef U3D_GMM_ALD40
constexpr bool GMM_SWZ = false;
constexpr int GMM_ALD = 40;
#else
constexpr bool GMM_SWZ = true;
constexpr int GMM_ALD = 32;
#endif

// workgroup-tile kernel (spconv_wg.hip): pr = 1 bf16 operands, 2 three bf16 planes, 3 bf16 source rows (p.src points at bf16 [n_src][Cs]); p.n_sub counts R-row tiles
int launch_gmm_wg(const GmmParams& p, int cs16, int R, int pr, hipStream_t s);
bool gmm_wg_supported(int cs16, int R, int pr);

}  // namespace u3d
