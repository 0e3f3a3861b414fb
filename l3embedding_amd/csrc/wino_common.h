// wino_common.h -- what the F(2x2,3x3) forward / data-gradient kernels share: the launch arguments and the output transform.
// conv_wino.hip (fp32 MFMA) and conv_wino_bx6.hip (split-bf16 operands on the bf16 MFMA) produce the same accumulator layout
// (32 x 32 C/D tiles: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)), so one epilogue serves both.
#pragma once
#include "kernels.h"
#include "device_common.h"

namespace l3 {

struct WinoArgs {
    const float* x;
    const float* u;
    const float* bias;
    float* y;
    int N, H, W, Cin, Cout;
    int TY, TX;        // 2x2 tiles per image
    int rows;          // N * TY flat tile rows
    int txb;           // tile-column blocks per row of tiles
    int mblocks, nblocks, nchunks;
    float inv_ty;      // 1 / TY: (n, ty) = divmod(flat tile row, TY) as one multiply (rows < 2^22)
    float* stat_part;  // SM != 0: per-(tile block) batch-norm partial sums [mblock][2][Cout] (bn_fused.hip layout)
    int stat_mode;     // SM == 1: 1 = moments of y, 2 = moments of relu(y) (the ReLU -> BN layer)
    BnBwdFuse bb;      // SM == 2: the launch is a data gradient; partials of the BatchNorm backward reduction (kernels.h)
    int dynamic = 0;       // ConvGeom::dynamic
    int* work = nullptr;   // persistent grid of conv_wino_kernel: its work counters (device_common.h wq_*), nullptr = static stride
};

struct TrueT { static constexpr bool value = true; };
struct FalseT { static constexpr bool value = false; };

// acc[i][jn]: tiles 32 i .. 32 i + 31 of the block (tile = tile row * BTX + tile column) x output channels n0 + 32 jn .. + 31;
// NW waves (16 or 8) took part: wave w accumulated positions w * PP .. w * PP + PP - 1 (PP = 16 / NW) in acc[pp], and transforms
// row pairs w * PP .. of every half.  Called by all waves after the last stage; leaves with the stage buffers dead.
template <int BTX, int SM, int NW = 16>
__device__ __forceinline__ void wino_output(const WinoArgs& a, f32x16 (&acc)[16 / NW][2][2], float* smem, int t, int wave, int lane,
                                            int R0, int tx0, int n0, int mb) {
    constexpr bool STATS = SM != 0;
    constexpr int PP = 16 / NW;
    const int l31 = lane & 31, half = lane >> 5;
    // ---- output transform: the 16 positions of one 32-tile x 64-channel half meet in LDS ------------------------
    // E[pos][row pair 16][col 64][2 rows]: a wave writes the two adjacent tile rows an accumulator register pair holds
    // as one ds_write_b64 per lane, and thread (wave = row pair, lane = column) reads its 16 positions of both rows
    // back as ds_read_b64 -- 85 and 256 B/clk against the 64 and 128 of 4-byte exchanges, and two rounds (four
    // barriers) instead of four.  Stores go through a buffer resource: the four pixels of a tile are one lane offset
    // plus scalar offsets, and "outside the image" is an out-of-range lane offset (dropped by the buffer unit).
    float* E = smem;
    const __amdgpu_buffer_rsrc_t ysrd =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (int)((size_t)a.N * a.H * a.W * a.Cout * 4), 0x00020000);
    const int so_x = a.Cout * 4, so_y = a.W * a.Cout * 4;
    const int ecol = lane;                                   // this thread's output channel n0 + ecol in the transform
    const float bz = a.bias != nullptr ? a.bias[n0 + ecol] : 0.f;
    // STATS: the BatchNorm that follows needs sum / sum of squares of this output per channel; take
    // them here, about the pivot bias[c] (the value the finalize kernel adds back), instead of
    // re-reading the tensor.
    float st0 = 0.f, st1 = 0.f;
    const bool srelu = SM == 1 && a.stat_mode == 2;
    // SM == 2: this output is dL/dy of a BatchNorm(+ReLU) with input bb.x: accumulate sum(d) and sum(d * x_hat),
    // d = the gradient where the forward ReLU let the value through, x_hat = (x - mean) * rstd
    float bsc = 0.f, bsh = 0.f, bmu = 0.f, brs = 0.f;
    __amdgpu_buffer_rsrc_t bxsrd = ysrd;
    if constexpr (SM == 2) {
        bxsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.bb.x, 0, (int)((size_t)a.N * a.H * a.W * a.Cout * 4), 0x00020000);
        bsc = a.bb.scale[n0 + ecol];
        bsh = a.bb.shift[n0 + ecol];
        bmu = a.bb.mean[n0 + ecol];
        brs = rsqrtf(a.bb.var[n0 + ecol] + a.bb.eps);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // the two tiles (rows 2 * wrow, 2 * wrow + 1 of this half, wrow = wave * PP + rp) this thread transforms, per row pair
        unsigned yv[PP][2][4];
#pragma unroll
        for (int rp = 0; rp < PP; ++rp)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int tbo = i * 32 + 2 * (wave * PP + rp) + e;
                const int rr = tbo / BTX, tc = tbo - rr * BTX;
                const int R = R0 + rr, tx = tx0 + tc;
                const int n = (int)(((float)R + 0.5f) * a.inv_ty), ty = R - n * a.TY;
                const int oy = 2 * ty, ox = 2 * tx;
                const bool ok = R < a.rows && tx < a.TX, okx = ox + 1 < a.W, oky = oy + 1 < a.H;
                const unsigned base = (unsigned)((((n * a.H + oy) * a.W + ox) * a.Cout + n0 + ecol) * 4);
                yv[rp][e][0] = ok ? base : 0x80000000u;
                yv[rp][e][1] = ok && okx ? base : 0x80000000u;
                yv[rp][e][2] = ok && oky ? base : 0x80000000u;
                yv[rp][e][3] = ok && okx && oky ? base : 0x80000000u;
            }
        float xl[PP][2][4];
#pragma unroll
        for (int rp = 0; rp < PP; ++rp)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int k = 0; k < 4; ++k) xl[rp][e][k] = 0.f;
        if constexpr (SM == 2) {       // the BatchNorm input at this thread's output pixels: in flight during the exchange
#pragma unroll
            for (int rp = 0; rp < PP; ++rp)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    xl[rp][e][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bxsrd, (int)yv[rp][e][0], 0, 0));
                    xl[rp][e][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bxsrd, (int)yv[rp][e][1], so_x, 0));
                    xl[rp][e][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bxsrd, (int)yv[rp][e][2], so_y, 0));
                    xl[rp][e][3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bxsrd, (int)yv[rp][e][3], so_x + so_y, 0));
                }
        }
#pragma unroll
        for (int pp = 0; pp < PP; ++pp)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {        // accumulator rows (r & 3) + 8 (r >> 2) + 4 half and the next one
                    const int pair = ((r & 3) >> 1) + 4 * (r >> 2) + 2 * half;
                    *reinterpret_cast<f32x2*>(E + (((wave * PP + pp) * 16 + pair) * 64 + jn * 32 + l31) * 2) =
                        f32x2{acc[pp][i][jn][r], acc[pp][i][jn][r + 1]};
                }
        __syncthreads();
#pragma unroll
        for (int rp = 0; rp < PP; ++rp) {
        f32x2 m2[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) m2[p] = *reinterpret_cast<const f32x2*>(E + ((p * 16 + wave * PP + rp) * 64 + ecol) * 2);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float s0[4], s1[4];
#pragma unroll
            for (int x4 = 0; x4 < 4; ++x4) {
                const float mid = m2[x4 * 4 + 1][e] + m2[x4 * 4 + 2][e], dif = m2[x4 * 4 + 1][e] - m2[x4 * 4 + 2][e];
                s0[x4] = m2[x4 * 4 + 0][e] + mid;
                s1[x4] = dif - m2[x4 * 4 + 3][e];
            }
            const float y00 = (s0[0] + bz) + (s0[1] + s0[2]), y01 = (s1[0] + bz) + (s1[1] + s1[2]);
            const float y10 = (s0[1] - s0[2]) + (bz - s0[3]), y11 = (s1[1] - s1[2]) + (bz - s1[3]);
            const float yy[4] = {y00, y01, y10, y11};
            if constexpr (SM == 2) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool pass = a.bb.relu != 1 || fmaf(xl[rp][e][k], bsc, bsh) > 0.f;
                    const float d = (int)yv[rp][e][k] >= 0 && pass ? yy[k] : 0.f;
                    st0 += d;
                    st1 = fmaf(d, (xl[rp][e][k] - bmu) * brs, st1);
                }
            }
            if constexpr (SM == 1) {
                const float pv = srelu ? fmaxf(bz, 0.f) : bz;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float val = srelu ? fmaxf(yy[k], 0.f) : yy[k];
                    const float d = (int)yv[rp][e][k] >= 0 ? val - pv : 0.f;
                    st0 += d;
                    st1 = fmaf(d, d, st1);
                }
            }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y00), ysrd, (int)yv[rp][e][0], 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y01), ysrd, (int)yv[rp][e][1], so_x, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y10), ysrd, (int)yv[rp][e][2], so_y, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y11), ysrd, (int)yv[rp][e][3], so_x + so_y, 0);
        }
        }
        __syncthreads();
    }
    if constexpr (STATS) {
        // red[which][wave][channel 64] -> one partial per (tile block, channel), the waves' sums added in order
        float* red = smem;
        red[(0 * NW + wave) * 64 + ecol] = st0;
        red[(1 * NW + wave) * 64 + ecol] = st1;
        __syncthreads();
        if (t < 128) {
            const int ch = t & 63, which = t >> 6;
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < NW; ++r) sum += red[(which * NW + r) * 64 + ch];
            a.stat_part[((size_t)mb * 2 + which) * a.Cout + n0 + ch] = sum;
        }
    }
}

}  // namespace l3
