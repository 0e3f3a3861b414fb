// wgw_common.h -- launch arguments and stage geometry shared by the Winograd F(3x3, 2x2) weight-gradient kernels:
// conv_wgrad_wino.hip (fp32 MFMA) and conv_wgrad_bx6.hip (split-bf16 operands on the bf16 MFMA).
#pragma once
#include "kernels.h"
#include "device_common.h"

namespace l3 {

struct WgwArgs {
    const float* x;       // (N, H, W, Cin)
    const float* dy;      // (N, H, W, Cout)
    float* part;          // [splits][16][Cin][Cout]
    int N, H, W, Cin, Cout;
    int uy, ux;           // units per image (rows, columns of UR x UC tile groups)
    int units;            // N * uy * ux
    int per_split, splits;
    int ctiles, ktiles;
};

template <int UC>
struct WgwGeom {
    static constexpr int UR = 8 / UC;
    static constexpr int XROWS = 2 * UR + 2, XPITCH = 2 * UC + 2, XPIX = XROWS * XPITCH;
    static constexpr int YPITCH = 2 * UC, YPIX = 32;
    static constexpr int XPIECES = (XPIX + 3) / 4;          // 1-KiB pieces = 4 pixels x 64 channels (18 / 15 / 15)
    static constexpr int YPIECES = YPIX / 4;
    static constexpr int PIECES = XPIECES + YPIECES;
    static constexpr int XBYTES = XPIECES * 1024;
    static constexpr int STAGE = PIECES * 1024;
    static constexpr size_t LDS_BYTES = 2 * (size_t)STAGE;
    static_assert(PIECES <= 32, "two pieces per wave at most");
    // tile q = 4 * half + j of the unit -> (tile row, tile column); split into the lane part and the immediate part
    __host__ __device__ static constexpr int lane_tr(int half) { return UC == 8 ? 0 : UC == 4 ? half : 2 * half; }
    __host__ __device__ static constexpr int lane_tc(int half) { return UC == 8 ? 4 * half : 0; }
    __host__ __device__ static constexpr int imm_tr(int j) { return UC == 2 ? j >> 1 : 0; }
    __host__ __device__ static constexpr int imm_tc(int j) { return UC == 2 ? j & 1 : j; }
};

// conv_wgrad_bx6.hip: the same partials from split-bf16 operands; stages of TWO units, so per_split must be even
void conv_wgrad_bx6_launch(const WgwArgs& a, int uc, hipStream_t s);

}  // namespace l3
